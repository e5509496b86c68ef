// effects.h -- host-side description of one effect (what the reference keeps in e->data) and the
// init-time parsers that produce it.  No device code here: a Spec is plain data that the pipeline
// compiler (engine.cpp) turns into fused GPU stages.
#pragma once
#include <array>
#include <memory>
#include <string>
#include <vector>
#include "dsp_effect_abi.h"
#include "host_util.h"

namespace dspamd {

enum class Kind { Gain, Add, Biquad, Remix, Mix, Crossfeed, Delay, Align, FirDirect, Conv, Resample };

enum ConvMode {
	CONV_ZERO_LATENCY = 0,   // fir_p (fir_p.c): y[n] = sum h[k] x[n-k], no added latency
	CONV_LATENCY_LEN = 1,    // fir (fir.c:109-149): same values, `latency` frames late, latency reported to the host
	CONV_ZITA_EQUIV = 2,     // zita_convolver contract: float32 in / filter / out, `latency` = part_len frames late
};

// one second-order section of a time-reversed IIR filter before the per-channel design (reverse_iir.c:40-60)
enum { RIIR_NONE = 0, RIIR_1R = 1, RIIR_2R = 2, RIIR_CC = 3 };
struct RiirSec {
	int pt = RIIR_NONE, qt = RIIR_NONE;      // kind of the pole pair / zero pair
	double pr[2] = { 0, 0 }, pc_re = 0, pc_im = 0;
	double qr[2] = { 0, 0 }, qc_re = 0, qc_im = 0;
	double rr[2] = { 0, 0 }, rc_re = 0, rc_im = 0;   // residues (filled by the design)
	double g = 1.0, thresh = 80.0;
};

struct Spec {
	Kind kind;
	std::string name;                        // effect name as the registry knows it (e->name)
	int fs_in = 0, fs_out = 0, ch_in = 0, ch_out = 0;
	Selector sel;                            // channels acted on (copy of channel_selector)
	int flags = 0;                           // EFFECT_FLAG_*

	std::vector<double> vec;                 // Gain/Add: per-channel operand (1.0 / 0.0 where unselected)
	std::vector<std::array<double, 5>> bq;   // Biquad: c0..c4 per channel (valid where sel)
	std::vector<Selector> remix;             // Remix: [ch_out] selectors over ch_in
	// Mix (st2ms / ms2st, st2ms.c:28-54; the two ends of an expanded crossfeed): out[k] = (sum_j in[idx[k][j]] * w[k][j]) * post[k],
	// products and sums single-rounded in list order -- `remix` holds the same dependencies as 0/1 rows for channel_deps
	std::vector<std::vector<int>> mix_idx;
	std::vector<std::vector<double>> mix_w;
	std::vector<double> mix_post;            // empty = no post-scale
	// Crossfeed (crossfeed.c:26-50): the pair (xf_c0, xf_c1), gains, first-order low-/high-pass sections
	int xf_c0 = -1, xf_c1 = -1;
	double xf_direct = 1.0, xf_cross = 0.0;
	std::array<double, 5> xf_lp{ { 1, 0, 0, 0, 0 } }, xf_hp{ { 1, 0, 0, 0, 0 } };
	std::vector<ssize_t> delay;              // Delay: requested per-channel delay (realised by Align); Align: ring length
	ssize_t discard = 0;                     // Align: leading frames dropped at end of chain (align.c:53-62)
	// Delay with a fractional part (`delay -f[order]`, delay.c:149-204): the amounts add up across merged effects
	// until prepare() splits them into an integer delay (host alignment) and a Thiran all-pass (a cascade section)
	std::vector<double> delay_frac;          // [ch_in] fractional amount still to be realised
	std::vector<int> fd_ap_n;                // [ch_in] all-pass order (0 = default) / after prepare: order in use
	bool frac_delay = false;                 // after prepare: a Biquad-kind spec that also requests delay[k] from the host
	// further sections of the same effect (all-pass orders above 2 factor into several second-order sections): [section][ch]
	std::vector<std::vector<std::array<double, 5>>> bq_more;
	std::vector<Selector> sel_more;

	std::vector<double> taps;                // FirDirect/Conv: [T][fch] interleaved
	int fch = 0;
	ssize_t T = 0, ref = 0, latency = 0;
	int conv_mode = CONV_ZERO_LATENCY;
	// reverse IIR (`biquad -r`, reverse_iir.c): per-channel section lists until prepare() turns them into one FIR per
	// channel (taps / T / fch) with its own delay, which the host compensates (req_delay[k] -= ch_latency[k])
	std::vector<std::vector<RiirSec>> riir;  // [ch_in]; non-empty = a reverse-IIR effect
	bool riir_pending = false;
	std::vector<ssize_t> ch_latency;         // [ch_in] when the channels' delays differ (reverse IIR)
	std::vector<std::string> riir_plot;      // [ch_in] transfer function of the designed channel in gnuplot notation (for e->plot)
	long max_part_len = 0;                   // fir_p: the host's partition-length cap -- only the number of terms `plot` prints depends on it

	int rs_n = 1, rs_d = 1, rs_m = 0;        // Resample: ratio n/d, prototype order m
	double rs_fc = 0.0;
	std::vector<double> rs_proto;            // windowed-sinc prototype s[0..m] at rate max(fs_in, fs_out) * sinc_os
	int rs_os = 1;
	std::vector<double> rs_pre;              // Resample: taps of a zero-latency FIR (fir_p) folded in front of the polyphase branches
	std::string rs_pre_name;
};

using SpecPtr = std::unique_ptr<Spec>;

// Parsers: argv[0] is the effect name; return nullptr after set_error() on any syntax/range error.
// *noop is set when the effect is valid but does nothing (host drops it: run == NULL).
SpecPtr parse_biquad(int effect_number, const stream_info *is, const char *sel, int argc, const char *const *argv, bool *unsupported_reverse);
SpecPtr parse_gain(int effect_number, const stream_info *is, const char *sel, int argc, const char *const *argv);
SpecPtr parse_remix(const stream_info *is, const char *sel, int argc, const char *const *argv);
SpecPtr parse_st2ms(int effect_number, const stream_info *is, const char *sel, int argc, const char *const *argv);
SpecPtr parse_crossfeed(const stream_info *is, const char *sel, int argc, const char *const *argv);
void crossfeed_expand(const Spec &xf, Spec &spread, Spec &filters, Spec &combine);   // the three device stages of one crossfeed
SpecPtr parse_delay(const stream_info *is, const char *sel, int argc, const char *const *argv, bool *noop);
SpecPtr parse_fir(const char *name, bool partitioned, const stream_info *is, const char *sel, const char *dir, int argc, const char *const *argv);
SpecPtr parse_zita(const stream_info *is, const char *sel, const char *dir, int argc, const char *const *argv);
SpecPtr parse_hilbert(const stream_info *is, const char *sel, int argc, const char *const *argv);
SpecPtr parse_resample(const stream_info *is, const char *sel, int argc, const char *const *argv, bool *noop);

SpecPtr make_fir_spec(const char *name, const stream_info *is, const char *sel, const double *filter, int fch, ssize_t T, ssize_t ref, int mode, int force_direct, int part_len);
SpecPtr make_delay_spec(const char *name, const stream_info *is, const char *sel, ssize_t samples, bool *noop);
SpecPtr make_align_spec(int fs, int channels, const std::vector<ssize_t> &len, ssize_t discard);

// merge(dest, src): the reference's e->merge callbacks (gain.c:57-79, biquad.c:344-376, delay.c:127-141)
bool merge_specs(Spec &dest, const Spec &src);

// design helpers (also used by tests through the C API)
void biquad_normalise(double b0, double b1, double b2, double a0, double a1, double a2, std::array<double, 5> &c);
void biquad_design(int type, double fs, double arg0, double arg1, double arg2, double arg3, int width_type, std::array<double, 5> &c);
void hilbert_design(ssize_t taps, double angle_rad, std::vector<double> &h);

// reverse_iir.cpp
void riir_sec_from_biquad(const std::array<double, 5> &c, double thresh, RiirSec *s);
bool riir_design(const char *name, int channel, std::vector<RiirSec> secs, std::vector<double> &taps, ssize_t *latency, std::string *plot = nullptr);
bool riir_prepare(Spec &sp);   // the effect's prepare(): section lists -> per-channel FIRs
SpecPtr make_frac_delay_spec(const char *name, const stream_info *is, const char *sel, double samples_frac, int fd_ap_n, bool *noop);
extern "C" __attribute__((visibility("hidden"))) int dspamd_thiran_pole_sections(int n, double D, double *out, int cap);   // thiran_roots.cpp (host compiler: 113-bit arithmetic); [cap][5] doubles out, sections written or -1
bool delay_prepare(Spec &sp, bool *noop);   // delay.c:149-204: integer part + Thiran all-pass section per channel

// number of filter terms the reference's fir_p prints in its plot (32 direct taps + the zero-padded partition groups, fir_p.c:242-289)
ssize_t fir_p_planned_len(ssize_t T, long max_part_len);

const effect_info *registry_lookup(const char *name);
const effect_info *registry_table(int *n);

}  // namespace dspamd
