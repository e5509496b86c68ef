// resample.cpp -- ResampleStage: host bookkeeping of the rational resampler (frames in / frames out, priming,
// drain2) around kernels_resample.hip.  Parity is defined on the concatenated stream and its total length
// (= ceil(N n / d)), not on per-call counts; per call at most ceil(frames n / d) frames are
// emitted, which is the bound the host sizes its buffers for (effects_chain.c:993-1002).
#include "stages.h"
#include "resample_params.h"
#include <cmath>
#include <cstdlib>
#include <sstream>

namespace dspamd {

static long mult_ceil(long v, int n, int d) { const long long r = (long long) v * n; return (long) ((r % d) ? r / d + 1 : r / d); }

class ResampleStage : public Stage {
public:
	bool init(const Spec &sp, ssize_t max_frames);
	const char *type() const override { return "resample"; }
	std::string describe() const override;
	ssize_t max_out_frames(ssize_t in_frames) const override { return mult_ceil(in_frames, n, d); }
	ssize_t run(const double *in, long in_stride, ssize_t frames, double *out, long out_stride, hipStream_t st) override;
	ssize_t drain2(ssize_t max_frames, double *out, long out_stride, hipStream_t st) override;
	void reset(hipStream_t st) override;
	size_t device_bytes() const override { return ring.bytes + tab.bytes + G.bytes; }
	// last stage of a pipeline run in wire formats: both kernels apply the sink sample by sample (any format)
	bool wire_out_ok(int fmt, const void *out, long out_stride, ssize_t frames, bool also_in, int in_fmt) const override
	{
		(void) fmt; (void) out; (void) out_stride; (void) frames; (void) in_fmt;
		return wire_fusion_on() && !also_in;
	}
private:
	ssize_t emit(long count, double *out, long out_stride, hipStream_t st);
	int n = 1, d = 1, J = 0, KT = 256;
	long out_delay = 0, ring_len = 0, q_total = 0, emitted = 0;
	DevBuf ring, tab, G;
	// GEMM form (kernels_resample.hip): blocks of NB = n g outputs <- DB = d g inputs
	bool gemm = false;
	int NB = 0, DB = 0, Kpad = 0, Npad = 0, log2cp = 0;
};

std::string ResampleStage::describe() const
{
	std::ostringstream o;
	o << (gemm ? "resample-gemm[" : "resample[") << fs_in << "->" << fs_out << " " << n << "/" << d << " taps/phase=" << J << " delay=" << out_delay << "]";
	return o.str();
}

static double albrecht(double x)
{
	static const double a[9] = {
		2.318028013590306028393e-1, 3.932575471789488615081e-1, 2.385434764970747429454e-1,
		1.014370437785239811268e-1, 2.911516061918003918645e-2, 5.280988177252078698806e-3,
		5.382909093381945363528e-4, 2.442086527507867730168e-5, 2.706153764205043532817e-7,
	};
	if (x >= 1.0 || x <= 0.0) return 0.0;
	double w = a[0];
	for (int i = 1; i < 9; ++i) w += ((i & 1) ? -a[i] : a[i]) * cos(2*i*M_PI*x);
	return w;
}

// tab[j][p] = A s_a((p + j n) os / min(n, d)), j < J, p < n: the polyphase taps of resample.c's frequency-domain
// resampler restated in the time domain; *out_delay as resample.c:312-316
void resample_polyphase_table(const Spec &sp, int *J_out, long *out_delay, std::vector<double> &t)
{
	const int n = sp.rs_n, d = sp.rs_d;
	const int m = sp.rs_m, os = sp.rs_os;
	const int mn = std::min(n, d);
	const int max_rate = std::max(sp.fs_in, sp.fs_out);
	const double fc_os = sp.rs_fc / os;
	const long m_os = (long) (m + 1) * os - 1;
	*out_delay = (sp.fs_out == max_rate) ? m / 2 : lround(m / 2 * ((double) n / d));
	const double A = (double) os * max_rate / sp.fs_in;
	const int J = (int) ceil((double) m_os * mn / ((double) os * n)) + 1;
	t.assign((size_t) J * n, 0.0);
	for (int j = 0; j < J; ++j) {
		for (int p = 0; p < n; ++p) {
			const double u = ((double) p + (double) j * n) * os / mn;
			const double x = (2.0 * u - m_os) / 2.0;
			const double sinc = (fabs(x) < 1e-9) ? fc_os : sin(M_PI*fc_os*x) / (M_PI*x);
			t[(size_t) j * n + p] = A * sinc * albrecht(u / m_os);
		}
	}
	*J_out = J;
}

bool ResampleStage::init(const Spec &sp, ssize_t max_frames)
{
	n = sp.rs_n;
	d = sp.rs_d;
	std::vector<double> t;
	resample_polyphase_table(sp, &J, &out_delay, t);
	if (!tab.upload(t.data(), t.size() * sizeof(double))) return false;
	long need = (long) J + std::max<long>(max_frames, 1) + (long) KT * d / n + 64;
	// GEMM form when the block fits the kernel's tiling (NB <= 192 columns, <= 8 channels, span in LDS)
	{
		const int g = (n >= 16) ? 1 : (16 + n - 1) / n;
		NB = n * g; DB = d * g;
		Npad = (NB + 31) & ~31;                 // two waves x PT column tiles, no guards in the kernel
		Kpad = (J + DB - 1 + 7) & ~7;          // the kernel walks K in steps of 8 and prefetches one step beyond (zero rows)
		log2cp = 0;
		while ((1 << log2cp) < ch_in) ++log2cp;
		gemm = !getenv("DSP_AMD_RESAMPLE_NO_GEMM") && Npad <= 192 && ch_in <= 8 && resample_gemm_lds_bytes(DB, J, log2cp) <= 160 * 1024;
		if (gemm) {
			std::vector<double> gm((size_t) (Kpad + 8) * Npad, 0.0);
			for (int r = 0; r < NB; ++r) {
				const long qr = ((long) r * d) / n;
				const int ph = (int) (((long) r * d) % n);
				for (int j = 0; j < J; ++j) {
					const long u = qr - j;                      // in [-(J-1), DB-1]
					gm[(size_t) (u + J - 1) * Npad + r] = t[(size_t) j * n + ph];
				}
			}
			if (!G.upload(gm.data(), gm.size() * sizeof(double))) return false;
			need += 2L * DB * (64 >> log2cp) + NB;
		}
	}
	ring_len = 1;
	while (ring_len < need) ring_len <<= 1;
	return ring.alloc((size_t) S * ring_len * ch_in * sizeof(double));
}

ssize_t ResampleStage::emit(long count, double *out, long out_stride, hipStream_t st)
{
	if (count <= 0) return 0;
	if (gemm) {
		ResampleGemmParams g;
		g.ring = ring.as<double>();
		g.ring_len = ring_len; g.ring_mask = ring_len - 1; g.q_total = q_total;
		g.G = G.as<double>();
		g.NB = NB; g.DB = DB; g.J = J; g.Kpad = Kpad; g.Npad = Npad;
		g.C = ch_in; g.log2cp = log2cp;
		g.out_delay = out_delay; g.m_first = emitted; g.m_count = count;
		g.i_first = (emitted + out_delay) / NB;
		g.out = out; g.out_stride_frames = out_stride; g.out_frame0 = 0;
		g.sink = wire_sink; g.sink_bs = (int) pcm_sample_bytes(wire_sink.fmt);
		{ ProfScope ps("resample_gemm_kernel", st); launch_resample_gemm(g, S, st); }
		emitted += count;
		return count;
	}
	ResampleParams p;
	p.ring = ring.as<double>();
	p.ring_len = ring_len; p.ring_mask = ring_len - 1;
	p.q_total = q_total;
	p.tab = tab.as<double>();
	p.n = n; p.d = d; p.J = J; p.C = ch_in; p.KT = KT;
	p.out_delay = out_delay;
	p.m_first = emitted; p.m_count = count;
	p.out = out; p.out_stride_frames = out_stride; p.out_frame0 = 0;
	p.sink = wire_sink; p.sink_bs = (int) pcm_sample_bytes(wire_sink.fmt);
	{ ProfScope ps("resample_kernel", st); launch_resample(p, S, st); }
	emitted += count;
	return count;
}

ssize_t ResampleStage::run(const double *in, long in_stride, ssize_t frames, double *out, long out_stride, hipStream_t st)
{
	{ ProfScope ps("resample_push_kernel", st); launch_resample_push(in, in_stride, ring.as<double>(), ring_len, q_total & (ring_len - 1), frames, ch_in, S, st); }
	q_total += frames;
	// full-rate outputs k with floor(k d / n) < q_total are computable: k < ceil(q_total n / d)
	const long avail = std::max<long>(0, mult_ceil(q_total, n, d) - out_delay) - emitted;
	const long cap = mult_ceil(frames, n, d);
	return emit(std::min(avail, cap), out, out_stride, st);
}

ssize_t ResampleStage::drain2(ssize_t max_frames, double *out, long out_stride, hipStream_t st)
{
	// total output length is ceil(N n / d) (resample.c:163-188): the tail is computed against zero input
	const long total = mult_ceil(q_total, n, d);
	const long left = total - emitted;
	if (q_total == 0 || left <= 0) return -1;
	const long cap = std::min<long>(mult_ceil(max_frames, n, d), (ring_len - J - 64) * n / d);
	return emit(std::min(left, std::max<long>(cap, 1)), out, out_stride, st);
}

void ResampleStage::reset(hipStream_t st)
{
	(void) hipMemsetAsync(ring.p, 0, ring.bytes, st);
	q_total = emitted = 0;
}

Stage *make_resample_stage(const Spec &sp, int n_streams, ssize_t max_frames)
{
	ResampleStage *s = new ResampleStage;
	s->S = n_streams; s->ch_in = sp.ch_in; s->ch_out = sp.ch_out; s->fs_in = sp.fs_in; s->fs_out = sp.fs_out;
	if (!s->init(sp, max_frames)) { delete s; return nullptr; }
	return s;
}

}  // namespace dspamd
