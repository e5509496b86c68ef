import os, sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import dsp_amd
def make_filter(n, seed=7, decay=8000.0):
    rng = np.random.default_rng(seed); h = rng.standard_normal(n) * np.exp(-np.arange(n) / decay); return h / np.sqrt(np.sum(h * h)) / 4.0
T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
h = make_filter(T); p = '/tmp/f.raw'; np.asarray(h, dtype='<f8').tofile(p)
chain = os.environ.get("PRE", "") + f" fir_p -t pcm -e double -c 1 {p} resample 96k"
S, C, N = int(os.environ.get("SS", "1")), int(os.environ.get("CC", "2")), 20000
x = np.random.Generator(np.random.PCG64(5)).uniform(-0.3, 0.3, size=(S, N, C))
def run():
    b = dsp_amd.BatchChain(chain, 48000, C, S, 8192)
    print(b.plan())
    return b.process(torch.from_numpy(x).cuda(), 8192).cpu().numpy()
ym = run()
os.environ['DSP_AMD_NO_LTI_MERGE'] = '1'
yu = run()
d = (ym - yu)[0]
print('shape', ym.shape, yu.shape, 'rms diff', np.sqrt(np.mean(d * d)), 'max', np.abs(d).max(), 'at', np.unravel_index(np.abs(d).argmax(), d.shape))
blk = 4096
for i in range(0, d.shape[0], blk * 4):
    print(i, np.sqrt(np.mean(d[i:i + blk * 4] ** 2)))
