# identical runs of the same batch must agree bit for bit (see rw_store_b128 in kernels_cascade.hip)
# usage: check_determinism.py <streams> <channels> <frames> [fir_p|fir <taps> [more effects ...]]
import sys, os, json, collections, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import dsp_amd, torch
chain = [c for c in json.load(open('tests/golden/golden.json'))['cases'] if c['name'] == 'config2'][0]['chain']
S, C, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if len(sys.argv) > 4:   # optional tail, e.g. 'fir_p 40000' / 'fir_p 40000 resample 96k': a seeded random filter of that many taps
    import tempfile
    taps = int(sys.argv[5]); rng = np.random.default_rng(1); h = rng.standard_normal(taps) * np.exp(-np.arange(taps) / 5000.0) / 50
    f = tempfile.NamedTemporaryFile(suffix='.raw', delete=False); h.astype('<f8').tofile(f); f.close()
    chain += f' {sys.argv[4]} -t pcm -e double -c 1 {f.name} ' + ' '.join(sys.argv[6:])
if os.environ.get('CHAIN'):   # whole chain from the environment instead
    chain = os.environ['CHAIN']
g = torch.Generator(device="cuda"); g.manual_seed(3)
x = torch.rand((S, n, C), dtype=torch.float64, device="cuda", generator=g) - 0.5
outs = []
for rep in range(4):
    b = dsp_amd.BatchChain(chain, 48000, C, S, n)
    outs.append(b.run(x).clone())
diff = sum(int((outs[0] != o).sum().item()) for o in outs[1:])
print("S", S, "C", C, "n", n, "elements differing between identical runs:", diff)
