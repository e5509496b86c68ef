// reverse_iir.cpp -- `biquad -r` and friends: time-reversed IIR sections (reverse_iir.c; M. Vicanek, "A New Reverse
// IIR Filtering Algorithm", 2015 / rev. 2022) for the GPU backend.
//
// At run time the reference evaluates, per channel, a PARALLEL form of truncated reversed one-pole responses: for
// every pole p with residue res a cascade of N comb stages y = p^(2^j) x + x[n - 2^j] (reverse_iir.c:92-105), i.e.
//     res * prod_{j<N} (p^(2^j) + z^(-2^j))  =  res * sum_{k < 2^N} p^(2^N - 1 - k) z^(-k),
// plus an FIR part delayed by 2^N samples (:107-125), repeated poles in further states run in series (:140-152).
// All of it is FEED-FORWARD: a channel's reverse-IIR effect IS an FIR filter of 2^N + fir.n (+ the series states)
// taps followed by nothing else, whose delay the host compensates through channel_offsets (:275-280).  This backend
// therefore designs that FIR on the host -- the pole/zero bookkeeping of reverse_iir_effect_prepare restated below, in the
// reference's own double / double-complex arithmetic where a rounding decides a stage power -- and hands it to the
// FFT convolver, one filter per channel.  Only the evaluation order of the feed-forward sums differs from the
// reference's comb cascade (fp64 rounding, ~1e-15).
#include "effects.h"
#include <cmath>
#include <complex>
#include <cfloat>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

namespace dspamd {

typedef std::complex<double> cd;

static const double POLE_CMP_TOL = 1e-4;                 // reverse_iir.c:36
static const double RES_LIM = 1e-8 / DBL_EPSILON;        // reverse_iir.c:37

static inline int pq_n(int t) { return t == RIIR_CC ? 2 : t; }           // reverse_iir.c:47
static inline int pq_n_eval(int t) { return t == RIIR_CC ? 1 : t; }      // reverse_iir.c:48

// The roots of 1 + p1 z^-1 + p2 z^-2 as the reference classifies them (reverse_iir.c:672-723: poles from (c3, c4), zeros from (c1, c2) / c0):
// none / one real root (p2 == 0), else a conjugate pair -- unless its imaginary part is below 1e-6, which counts as two real roots of a
// discriminant clamped at zero.  The arithmetic (which quotient is formed first, where the clamp sits) decides which sections a chain splits
// into, so it is the reference's; the shape of the code is not.
struct QuadRoots { int type = RIIR_NONE; double re[2] = { 0.0, 0.0 }; cd pair = cd(0.0, 0.0); };
static QuadRoots quad_roots(double p1, double p2)
{
	QuadRoots q;
	if (p2 == 0.0) {
		if (p1 != 0.0) { q.type = RIIR_1R; q.re[0] = -p1; }
		return q;
	}
	const double disc = p1 * p1 - 4.0 * p2;
	if (disc < 0.0) {
		const cd root = (std::sqrt(cd(disc, 0.0)) - p1) / 2.0;
		if (std::fabs(root.imag()) >= 1e-6) { q.type = RIIR_CC; q.pair = root; return q; }
	}
	const double sq = std::sqrt(disc > 0.0 ? disc : 0.0);
	q.type = RIIR_2R;
	q.re[0] = (sq - p1) / 2.0;
	q.re[1] = (-sq - p1) / 2.0;
	return q;
}

void riir_sec_from_biquad(const std::array<double, 5> &c, double thresh, RiirSec *s)
{
	*s = RiirSec();
	s->thresh = thresh;
	s->g = c[0];
	const QuadRoots poles = quad_roots(c[3], c[4]);
	const QuadRoots zeros = (c[2] == 0.0 && c[1] == 0.0) ? QuadRoots() : quad_roots(c[1] / c[0], c[2] / c[0]);
	s->pt = poles.type; s->pr[0] = poles.re[0]; s->pr[1] = poles.re[1]; s->pc_re = poles.pair.real(); s->pc_im = poles.pair.imag();
	s->qt = zeros.type; s->qr[0] = zeros.re[0]; s->qr[1] = zeros.re[1]; s->qc_re = zeros.pair.real(); s->qc_im = zeros.pair.imag();
}

// one factor (z - root_i) / z of a section's numerator or denominator at z (reverse_iir.c:342-355)
static cd eval_pq(const double r[2], cd c, int type, int i, cd z)
{
	switch (type) {
	case RIIR_CC: return (z - (i ? std::conj(c) : c)) / z;
	case RIIR_2R: return (z - r[i ? 1 : 0]) / z;
	case RIIR_1R: return i ? cd(1.0, 0.0) : (z - r[0]) / z;
	default: return cd(1.0, 0.0);
	}
}
static cd sec_pc(const RiirSec &s) { return cd(s.pc_re, s.pc_im); }
static cd sec_qc(const RiirSec &s) { return cd(s.qc_re, s.qc_im); }

// does any pole of s0 coincide (within tolerance) with a pole of s1?  (reverse_iir.c:357-376)
static bool poles_close(const RiirSec &s0, const RiirSec &s1)
{
	for (int i = 0; i < pq_n(s0.pt); ++i) {
		switch (s1.pt) {
		case RIIR_CC:
			if (std::abs(eval_pq(s0.pr, sec_pc(s0), s0.pt, i, sec_pc(s1))) < POLE_CMP_TOL) return true;
			break;
		case RIIR_2R:
			if (std::abs(eval_pq(s0.pr, sec_pc(s0), s0.pt, i, cd(s1.pr[1], 0.0))) < POLE_CMP_TOL) return true;
			/* fallthrough */
		case RIIR_1R:
			if (std::abs(eval_pq(s0.pr, sec_pc(s0), s0.pt, i, cd(s1.pr[0], 0.0))) < POLE_CMP_TOL) return true;
			break;
		default: break;
		}
	}
	return false;
}

static long min_stages(double thresh, double abs_p)       // reverse_iir.c:378
{
	return lrint(ceil(log2(-(thresh + 6.02) / (20.0 * log10(abs_p)))));
}

// denominator / numerator polynomial coefficients of a section (reverse_iir.c:321-340)
static void expand_pq(const double r[2], cd c, int type, double out[2])
{
	out[0] = out[1] = 0.0;
	switch (type) {
	case RIIR_CC: out[0] = -2.0 * c.real(); out[1] = (c * std::conj(c)).real(); break;
	case RIIR_2R: out[0] = -r[0] - r[1]; out[1] = r[0] * r[1]; break;
	case RIIR_1R: out[0] = -r[0]; break;
	default: break;
	}
}

// squaring the way the reference's `p2n *= p2n` rounds it (double complex product without contraction)
static cd csquare(cd a)
{
	volatile double xx = a.real() * a.real(), yy = a.imag() * a.imag(), xy = a.real() * a.imag();
	volatile double re = xx - yy, im = xy + xy;
	return cd(re, im);
}

struct RiirStateFir { std::vector<long double> h; };    // equivalent FIR of one riir_state

// The reference's per-channel prepare (reverse_iir.c:381-636), producing the equivalent FIR instead of comb stages.
static void appendf(std::string &s, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
static void appendf(std::string &s, const char *fmt, ...)
{
	char buf[256];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	s += buf;
}

bool riir_design(const char *name, int channel, std::vector<RiirSec> v, std::vector<double> &taps, ssize_t *latency, std::string *plot)
{
	std::vector<RiirSec> cascade;
	// sections with a repeated real pole are split, the second copy goes to a series state (:394-411)
	for (size_t i = 0; i < v.size(); ++i) {
		RiirSec &sec = v[i];
		if (sec.pt == RIIR_2R && std::fabs(sec.pr[1] - sec.pr[0]) < POLE_CMP_TOL) {
			RiirSec split;
			split.thresh = sec.thresh;
			split.pt = sec.pt = RIIR_1R;
			split.pr[0] = sec.pr[1];
			if (sec.qt == RIIR_2R) {
				split.qt = sec.qt = RIIR_1R;
				split.qr[0] = sec.qr[1];
				split.g = sec.g = std::sqrt(sec.g);
			}
			else split.g = 1.0;
			cascade.push_back(split);
		}
	}
	std::vector<RiirStateFir> states;
	*latency = 0;
	for (;;) {
		// any other repeated pole moves to the next series state (:413-424)
		for (size_t i = 0; i < v.size(); ++i) {
			if (v[i].pt == RIIR_NONE) continue;
			for (size_t j = i + 1; j < v.size();) {
				if (poles_close(v[i], v[j])) { cascade.push_back(v[j]); v.erase(v.begin() + j); }
				else ++j;
			}
		}
		// number of comb stages, pole / zero counts (:426-455)
		long N = 3;
		int nq = 0, np = 0;
		double g = 1.0;
		for (const RiirSec &sec : v) {
			nq += pq_n(sec.qt);
			np += pq_n(sec.pt);
			g *= sec.g;
			long p0N = 0, p1N = 0;
			switch (sec.pt) {
			case RIIR_CC: p0N = min_stages(sec.thresh, std::abs(sec_pc(sec))); break;
			case RIIR_2R: p1N = min_stages(sec.thresh, std::fabs(sec.pr[1])); /* fallthrough */
			case RIIR_1R: p0N = min_stages(sec.thresh, std::fabs(sec.pr[0])); break;
			default: break;
			}
			N = std::max(N, std::max(p0N, p1N));
		}
		if (nq - np + 1 > 8) { set_error("%s: error: channel %d: too many zeros: %d-%d+1 > 8", name, channel, nq, np); return false; }
		if (N > 22) { set_error("%s: error: channel %d: reverse IIR needs 2^%ld taps (pole too close to the unit circle)", name, channel, N); return false; }
		// partial fraction residues (:457-483)
		bool do_cascade = false;
		for (RiirSec &sec : v) {
			const bool is_cc = (sec.pt == RIIR_CC);
			for (int l = 0; l < pq_n_eval(sec.pt); ++l) {
				const cd p = is_cc ? sec_pc(sec) : cd(sec.pr[l], 0.0);
				// z^(nq-np+1) (1 - p z^-1) H(z) at z = p
				cd num = (nq < np) ? cd(1.0, 0.0) : (nq == np) ? p : std::pow(p, nq - np + 1), den(1.0, 0.0);
				for (const RiirSec &es : v) {
					num *= eval_pq(es.qr, sec_qc(es), es.qt, 0, p);
					num *= eval_pq(es.qr, sec_qc(es), es.qt, 1, p);
					if (&es != &sec) den *= eval_pq(es.pr, sec_pc(es), es.pt, l, p);
					den *= eval_pq(es.pr, sec_pc(es), es.pt, l ? 0 : 1, p);
				}
				cd res = num / den;
				if (std::isnan(std::abs(res))) res = cd(HUGE_VAL, 0.0);
				if (std::abs(res) > RES_LIM) do_cascade = true;
				if (is_cc) { const cd r = g * res; sec.rc_re = r.real(); sec.rc_im = r.imag(); }
				else sec.rr[l] = g * res.real();
			}
		}
		if (do_cascade) {
			// ill-conditioned expansion: the section with the largest residue moves to the next series state (:485-504)
			if (v.size() < 2) { set_error("%s: error: reverse IIR partial fraction expansion failed", name); return false; }
			auto res_abs = [](const RiirSec &sec) {
				switch (sec.pt) {
				case RIIR_CC: return std::abs(cd(sec.rc_re, sec.rc_im));
				case RIIR_2R: return std::max(std::fabs(sec.rr[0]), std::fabs(sec.rr[1]));
				case RIIR_1R: return std::fabs(sec.rr[0]);
				default: return 0.0;
				}
			};
			size_t rm = 0;
			double max_res = res_abs(v[0]);
			for (size_t i = 1; i < v.size(); ++i) {
				const double r = res_abs(v[i]);
				if (r > max_res) { rm = i; max_res = r; }
			}
			cascade.push_back(v[rm]);
			v.erase(v.begin() + rm);
			continue;
		}
		// (the reference sorts the sections to minimise the run-time quantisation error, :506-533; the order of a
		// feed-forward sum does not change the FIR it implements)

		// FIR part when there are at least as many zeros as poles (:535-559)
		int fir_n = 0;
		double fir_c[8] = { 0 };
		if (nq >= np) {
			fir_n = nq - np + 1;
			fir_c[nq - np] = g;
			if (nq > np) {
				for (const RiirSec &sec : v) {
					double b[2], a[2];
					expand_pq(sec.qr, sec_qc(sec), sec.qt, b);
					expand_pq(sec.pr, sec_pc(sec), sec.pt, a);
					// biquad_init(&bq, 1, b0, b1, 1, a0, a1); c[n] = biquad(&bq, c[n]) for n = nq-np .. 0  (biquad.h:76-92)
					double m0 = 0.0, m1 = 0.0;
					for (int n = nq - np; n >= 0; --n) {
						const double s = fir_c[n];
						const double r = 1.0 * s + m0;
						m0 = m1 + b[0] * s - a[0] * r;
						m1 = b[1] * s - a[1] * r;
						fir_c[n] = r;
					}
				}
			}
		}
		// equivalent FIR of this state: sum over poles of res * p^(2^N - 1 - k), k < 2^N, with the stage powers p^(2^j)
		// squared in double as the reference does (:379-392 INIT_FILTER_STAGES), plus the FIR part behind a 2^N delay
		const long L = 1L << N;
		RiirStateFir st;
		st.h.assign((size_t) L + (fir_n > 0 ? fir_n : 0), 0.0L);
		auto add_pole = [&](cd p, cd res, bool conj_pair) {
			std::vector<cd> a((size_t) N);
			a[0] = p;
			for (long j = 1; j < N; ++j) a[j] = csquare(a[j - 1]);
			// p^m for m < 2^N as the product of the stage powers selected by the bits of m
			std::vector<std::complex<long double>> pw((size_t) L);
			pw[0] = std::complex<long double>(1.0L, 0.0L);
			for (long j = 0; j < N; ++j) {
				const std::complex<long double> aj((long double) a[j].real(), (long double) a[j].imag());
				const long half = 1L << j;
				for (long m = 0; m < half; ++m) pw[half + m] = pw[m] * aj;
			}
			const std::complex<long double> r((long double) res.real(), (long double) res.imag());
			for (long k = 0; k < L; ++k) {
				const std::complex<long double> t = r * pw[L - 1 - k];
				st.h[k] += conj_pair ? 2.0L * t.real() : t.real();
			}
		};
		for (const RiirSec &sec : v) {
			if (sec.pt == RIIR_CC) add_pole(sec_pc(sec), cd(sec.rc_re, sec.rc_im), true);
			else for (int j = 0; j < pq_n_eval(sec.pt); ++j) add_pole(cd(sec.pr[j], 0.0), cd(sec.rr[j], 0.0), false);
		}
		for (int m = 0; m < fir_n; ++m) st.h[(size_t) L + m] += (long double) fir_c[m];
		if (fir_n == 0) st.h.resize((size_t) L);
		states.push_back(std::move(st));
		if (plot) {
			// the transfer function of this series state in the reference's gnuplot notation (reverse_iir.c:178-212): the FIR
			// part behind its 2^N delay, then every pole as residue * prod_j (p^(2^j) + z^(-2^j))
			std::string &o = *plot;
			o += "*(0";
			if (fir_n > 0) {
				appendf(o, "+(%.15e", fir_c[0]);
				for (int m = 1; m < fir_n; ++m) appendf(o, "+%.15e*exp(-%d*j*w)", fir_c[m], m);
				appendf(o, ")*exp(-2**%ld*j*w)", N);
			}
			for (const RiirSec &sec : v) {
				if (sec.pt == RIIR_CC) {
					for (int half = 0; half < 2; ++half) {
						const double sg = half ? -1.0 : 1.0;
						appendf(o, "+{%.15e,%.15e}", sec.rc_re, sg * sec.rc_im);
						for (long jj = 0; jj < N; ++jj) appendf(o, "*({%.15e,%.15e}**(2**%ld)+exp(-2**%ld*j*w))", sec.pc_re, sg * sec.pc_im, jj, jj);
					}
				}
				else for (int l = 0; l < pq_n_eval(sec.pt); ++l) {
					appendf(o, "+%.15e", sec.rr[l]);
					for (long jj = 0; jj < N; ++jj) appendf(o, "*((%.15e)**(2**%ld)+exp(-2**%ld*j*w))", sec.pr[l], jj, jj);
				}
			}
			o += ")";
		}
		*latency += L + fir_n - 1;                                 // reverse_iir.c:623-625
		if (cascade.empty()) break;
		v.swap(cascade);
		cascade.clear();
	}
	// series states = convolution of their FIRs
	std::vector<long double> h = states[0].h;
	for (size_t s = 1; s < states.size(); ++s) {
		const std::vector<long double> &b = states[s].h;
		std::vector<long double> c(h.size() + b.size() - 1, 0.0L);
		for (size_t i = 0; i < h.size(); ++i) {
			if (h[i] == 0.0L) continue;
			for (size_t j = 0; j < b.size(); ++j) c[i + j] += h[i] * b[j];
		}
		h.swap(c);
	}
	taps.resize(h.size());
	for (size_t i = 0; i < h.size(); ++i) taps[i] = (double) h[i];
	return true;
}

}  // namespace dspamd
