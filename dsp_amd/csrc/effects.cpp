// effects.cpp -- init-time parsing and filter design (host only; libm on the host so that
// coefficients agree with the reference's to the last bit -- SURVEY.md section 8 a4).
#include "effects.h"
#include <dlfcn.h>
#include <cfloat>
#include <cerrno>
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>

namespace dspamd {

static Selector copy_sel(const char *sel, int n)
{
	return Selector(sel, sel + n);
}

static SpecPtr new_spec(Kind k, const char *name, const stream_info *is, const char *sel)
{
	SpecPtr s(new Spec);
	s->kind = k;
	s->name = name;
	s->fs_in = s->fs_out = is->fs;
	s->ch_in = s->ch_out = is->channels;
	s->sel = copy_sel(sel, is->channels);
	return s;
}

static void usage(const char *name)
{
	const effect_info *ei = registry_lookup(name);
	if (ei) log_msg(LL_ERROR, "%s: usage: %s %s", ei->name, ei->name, ei->usage);
}

// ------------------------------------------------------------------ biquad

enum { W_Q = 1, W_SLOPE, W_SLOPE_DB, W_BW_OCT, W_BW_HZ };   // biquad.h:53-59

void biquad_normalise(double b0, double b1, double b2, double a0, double a1, double a2, std::array<double, 5> &c)
{
	c = { b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0 };       // biquad.c:91-99
}

// Audio-EQ-cookbook designs; argument conventions of biquad.c:111-294
void biquad_design(int type, double fs, double arg0, double arg1, double arg2, double arg3, int width_type, std::array<double, 5> &c)
{
	struct { double b0 = 1, b1 = 0, b2 = 0, a0 = 1, a1 = 0, a2 = 0; } q;
	if (type == DSPAMD_BIQUAD_LOWPASS_TRANSFORM || type == DSPAMD_BIQUAD_HIGHPASS_TRANSFORM) {
		// Linkwitz transform: zeros cancel the existing (fz, qz) response, poles put (fp, qp) in its place
		const bool lp = (type == DSPAMD_BIQUAD_LOWPASS_TRANSFORM);
		const double w0z = 2*M_PI*arg0 / fs, w0p = 2*M_PI*arg2 / fs;
		const double cz = cos(w0z), cp = cos(w0p);
		const double az = sin(w0z) / (2.0*arg1), ap = sin(w0p) / (2.0*arg3);
		const double kz = lp ? 2.0/(1.0-cz) : 2.0/(1.0+cz);
		const double kp = lp ? 2.0/(1.0-cp) : 2.0/(1.0+cp);
		q.b0 = (1.0 + az)*kz; q.b1 = (-2.0 * cz)*kz; q.b2 = (1.0 - az)*kz;
		q.a0 = (1.0 + ap)*kp; q.a1 = (-2.0 * cp)*kp; q.a2 = (1.0 - ap)*kp;
		biquad_normalise(q.b0, q.b1, q.b2, q.a0, q.a1, q.a2, c);
		return;
	}
	double f0 = arg0, width = arg1;
	const double gain = arg2;
	if (width_type == W_SLOPE_DB) {
		width_type = W_SLOPE;
		width /= 12.0;
		if (type == DSPAMD_BIQUAD_LOWSHELF) f0 *= pow(10.0, fabs(gain) / 80.0 / width);
		else if (type == DSPAMD_BIQUAD_HIGHSHELF) f0 /= pow(10.0, fabs(gain) / 80.0 / width);
	}
	const double A = pow(10.0, gain / 40.0);
	const double w0 = 2*M_PI*f0 / fs;
	const double sn = sin(w0), cs = cos(w0);
	double alpha;
	if (width_type == W_SLOPE) alpha = sn/2.0 * sqrt((A + 1.0/A) * (1.0/width - 1.0) + 2.0);
	else if (width_type == W_BW_OCT) alpha = sn * sinh(M_LN2/2 * width * w0 / sn);
	else if (width_type == W_BW_HZ) alpha = sn / (2.0 * f0 / width);
	else alpha = sn / (2.0 * width);
	const double k1 = 1.0 + cs;   // first-order helpers
	switch (type) {
	case DSPAMD_BIQUAD_LOWPASS_1:   q.b0 = sn; q.b1 = sn; q.a0 = sn + k1; q.a1 = sn - k1; break;
	case DSPAMD_BIQUAD_HIGHPASS_1:  q.b0 = k1; q.b1 = -k1; q.a0 = sn + k1; q.a1 = sn - k1; break;
	case DSPAMD_BIQUAD_ALLPASS_1:   q.b0 = sn - k1; q.b1 = sn + k1; q.a0 = q.b1; q.a1 = q.b0; break;
	case DSPAMD_BIQUAD_LOWSHELF_1:  q.b0 = A*sn + k1; q.b1 = A*sn - k1; q.a0 = sn/A + k1; q.a1 = sn/A - k1; break;
	case DSPAMD_BIQUAD_HIGHSHELF_1: q.b0 = sn + k1*A; q.b1 = sn - k1*A; q.a0 = sn + k1/A; q.a1 = sn - k1/A; break;
	case DSPAMD_BIQUAD_LOWPASS_1P: {
		const double t = 1.0 - cs;
		q.b0 = -t + sqrt(t*t + 2.0*t); q.b1 = 0.0; q.a0 = 1.0; q.a1 = -1.0 + q.b0;
		break;
	}
	case DSPAMD_BIQUAD_LOWPASS:
		q.b0 = (1.0 - cs) / 2.0; q.b1 = 1.0 - cs; q.b2 = q.b0;
		q.a0 = 1.0 + alpha; q.a1 = -2.0*cs; q.a2 = 1.0 - alpha; break;
	case DSPAMD_BIQUAD_HIGHPASS:
		q.b0 = (1.0 + cs) / 2.0; q.b1 = -(1.0 + cs); q.b2 = q.b0;
		q.a0 = 1.0 + alpha; q.a1 = -2.0*cs; q.a2 = 1.0 - alpha; break;
	case DSPAMD_BIQUAD_BANDPASS_SKIRT:
		q.b0 = sn / 2.0; q.b1 = 0.0; q.b2 = -q.b0;
		q.a0 = 1.0 + alpha; q.a1 = -2.0*cs; q.a2 = 1.0 - alpha; break;
	case DSPAMD_BIQUAD_BANDPASS_PEAK:
		q.b0 = alpha; q.b1 = 0.0; q.b2 = -alpha;
		q.a0 = 1.0 + alpha; q.a1 = -2.0*cs; q.a2 = 1.0 - alpha; break;
	case DSPAMD_BIQUAD_NOTCH:
		q.b0 = 1.0; q.b1 = -2.0*cs; q.b2 = 1.0;
		q.a0 = 1.0 + alpha; q.a1 = q.b1; q.a2 = 1.0 - alpha; break;
	case DSPAMD_BIQUAD_ALLPASS:
		q.b0 = 1.0 - alpha; q.b1 = -2.0*cs; q.b2 = 1.0 + alpha;
		q.a0 = q.b2; q.a1 = q.b1; q.a2 = q.b0; break;
	case DSPAMD_BIQUAD_PEAK:
		q.b0 = 1.0 + alpha*A; q.b1 = -2.0*cs; q.b2 = 1.0 - alpha*A;
		q.a0 = 1.0 + alpha/A; q.a1 = q.b1; q.a2 = 1.0 - alpha/A; break;
	case DSPAMD_BIQUAD_LOWSHELF: {
		const double t = 2.0 * sqrt(A) * alpha;
		q.b0 = A * ((A + 1.0) - (A - 1.0)*cs + t);
		q.b1 = 2.0 * A * ((A - 1.0) - (A + 1.0)*cs);
		q.b2 = A * ((A + 1.0) - (A - 1.0)*cs - t);
		q.a0 = (A + 1.0) + (A - 1.0)*cs + t;
		q.a1 = -2.0 * ((A - 1.0) + (A + 1.0)*cs);
		q.a2 = (A + 1.0) + (A - 1.0)*cs - t;
		break;
	}
	case DSPAMD_BIQUAD_HIGHSHELF: {
		const double t = 2.0 * sqrt(A) * alpha;
		q.b0 = A * ((A + 1.0) + (A - 1.0)*cs + t);
		q.b1 = -2.0 * A * ((A - 1.0) + (A + 1.0)*cs);
		q.b2 = A * ((A + 1.0) + (A - 1.0)*cs - t);
		q.a0 = (A + 1.0) - (A - 1.0)*cs + t;
		q.a1 = 2.0 * ((A - 1.0) - (A + 1.0)*cs);
		q.a2 = (A + 1.0) - (A - 1.0)*cs - t;
		break;
	}
	}
	biquad_normalise(q.b0, q.b1, q.b2, q.a0, q.a1, q.a2, c);
}

// biquad.c:27-89: "<number>[q|s|d|o|h|k]" or "bw<order>[.<index>]"
static double parse_width(const char *s, int *type, char **endptr)
{
	*type = W_Q;
	double w = M_SQRT1_2;
	if (s[0] == 'b' && s[1] == 'w' && s[2] != '\0') {
		const char *p = s + 2;
		const long order = strtol(p, endptr, 10);
		if (*endptr == p || (**endptr != '\0' && **endptr != '.')) goto fail;
		if (order < 2) { set_error("parse_width(): filter order must be >= 2"); goto fail; }
		{
			const int nb = (int) (order / 2);
			long idx = 0;
			if (**endptr == '.') {
				p = *endptr + 1;
				idx = strtol(p, endptr, 10);
				if (*endptr == p || **endptr != '\0') goto fail;
				if (idx < 0 || idx >= nb) { set_error("parse_width(): filter index out of range"); goto fail; }
			}
			idx = nb - idx;
			w = 1.0 / (2.0 * sin(M_PI / order * (idx - 0.5)));
		}
		return w;
	}
	w = strtod(s, endptr);
	if (*endptr != s) {
		switch (**endptr) {
		case 'q': *type = W_Q; ++*endptr; break;
		case 's': *type = W_SLOPE; ++*endptr; break;
		case 'd': *type = W_SLOPE_DB; ++*endptr; break;
		case 'o': *type = W_BW_OCT; ++*endptr; break;
		case 'k': w *= 1000.0;  // fall through
		case 'h': *type = W_BW_HZ; ++*endptr; break;
		}
		if (**endptr != '\0') log_msg(LL_ERROR, "parse_width(): trailing characters: %s", *endptr);     // biquad.c:82
	}
	return w;
fail:
	*endptr = const_cast<char *>(s);
	return w;
}

SpecPtr parse_biquad(int num, const stream_info *is, const char *sel, int argc, const char *const *argv, bool *reverse)
{
	const char *name = argv[0];
	GetOpt g;
	int opt;
	*reverse = false;
	static const int n_args_of[] = { 0, 1, 1, 1, 2, 2, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 4, 4, 0, 6 };
	if (num < 1 || num > DSPAMD_BIQUAD_BIQUAD) { set_error("%s: BUG: unknown effect number %d", name, num); return nullptr; }
	const int n_args = n_args_of[num];
	double thresh = 80.0;                                   // biquad.c:450
	while ((opt = g.next(argc - n_args, argv, "r::")) != -1) {
		if (opt == 'r') {
			*reverse = true;
			if (g.arg) {                                    // biquad.c:391-395
				char *e2;
				thresh = (double) strtol(g.arg, &e2, 10);
				if (bad_endptr(name, g.arg, e2, "thresh")) { usage(name); return nullptr; }
				if (!(thresh >= 10.0 && thresh <= 200.0)) { set_error("%s: error: parameter out of range: thresh", name); usage(name); return nullptr; }   // (option errors print the usage: biquad.c:394, 461-464)
			}
		}
		else { g.print_error(opt, name); usage(name); return nullptr; }
	}
	if (argc - g.ind != n_args) { usage(name); return nullptr; }
	const char *const *a = argv + g.ind;
	char *end;
	int wt = W_Q, type = num;
	double v[4] = { 0, 0, 0, 0 };
	std::array<double, 5> c;
	auto freq = [&](const char *s, const char *what, double &out) {
		out = parse_freq(s, &end);
		if (bad_endptr(name, s, end, what)) return false;
		if (!(out >= 0.0 && out < is->fs / 2.0)) { set_error("%s: error: parameter out of range: %s", name, what); return false; }
		return true;
	};
	auto num_arg = [&](const char *s, const char *what, double &out) {
		out = strtod(s, &end);
		return !bad_endptr(name, s, end, what);
	};
	auto width = [&](const char *s, const char *what, double &out) {
		out = parse_width(s, &wt, &end);
		if (bad_endptr(name, s, end, what)) return false;
		if (!(out > 0.0)) { set_error("%s: error: parameter out of range: %s", name, what); return false; }
		return true;
	};
	const bool slope = (wt == W_SLOPE || wt == W_SLOPE_DB);
	(void) slope;
	switch (num) {
	case DSPAMD_BIQUAD_LOWPASS_1: case DSPAMD_BIQUAD_HIGHPASS_1: case DSPAMD_BIQUAD_ALLPASS_1: case DSPAMD_BIQUAD_LOWPASS_1P:
		if (!freq(a[0], "f0", v[0])) return nullptr;
		break;
	case DSPAMD_BIQUAD_LOWSHELF_1: case DSPAMD_BIQUAD_HIGHSHELF_1:
		if (!freq(a[0], "f0", v[0]) || !num_arg(a[1], "gain", v[2])) return nullptr;
		break;
	case DSPAMD_BIQUAD_LOWPASS: case DSPAMD_BIQUAD_HIGHPASS: case DSPAMD_BIQUAD_BANDPASS_SKIRT:
	case DSPAMD_BIQUAD_BANDPASS_PEAK: case DSPAMD_BIQUAD_NOTCH: case DSPAMD_BIQUAD_ALLPASS:
		if (!freq(a[0], "f0", v[0]) || !width(a[1], "width", v[1])) return nullptr;
		if (wt == W_SLOPE || wt == W_SLOPE_DB) { set_error("%s: error: invalid width type", name); return nullptr; }
		break;
	case DSPAMD_BIQUAD_PEAK: case DSPAMD_BIQUAD_LOWSHELF: case DSPAMD_BIQUAD_HIGHSHELF:
		if (!freq(a[0], "f0", v[0]) || !width(a[1], "width", v[1])) return nullptr;
		if (num == DSPAMD_BIQUAD_PEAK && (wt == W_SLOPE || wt == W_SLOPE_DB)) { set_error("%s: error: invalid width type", name); return nullptr; }
		if (!num_arg(a[2], "gain", v[2])) return nullptr;
		break;
	case DSPAMD_BIQUAD_LOWPASS_TRANSFORM: case DSPAMD_BIQUAD_HIGHPASS_TRANSFORM:
		if (!freq(a[0], "fz", v[0]) || !width(a[1], "width_z", v[1])) return nullptr;
		if (wt != W_Q) { set_error("%s: error: invalid width type", name); return nullptr; }
		if (!freq(a[2], "fp", v[2]) || !width(a[3], "width_p", v[3])) return nullptr;
		if (wt != W_Q) { set_error("%s: error: invalid width type", name); return nullptr; }
		break;
	case DSPAMD_BIQUAD_DEEMPH:  // CD de-emphasis presets, biquad.c:503-521
		type = DSPAMD_BIQUAD_HIGHSHELF;
		wt = W_SLOPE;
		if (is->fs == 44100) { v[0] = 5283; v[1] = 0.4845; v[2] = -9.477; }
		else if (is->fs == 48000) { v[0] = 5356; v[1] = 0.479; v[2] = -9.62; }
		else { set_error("%s: error: sample rate must be 44100 or 48000", name); return nullptr; }
		break;
	case DSPAMD_BIQUAD_BIQUAD: {
		double r[6];
		static const char *nm[6] = { "b0", "b1", "b2", "a0", "a1", "a2" };
		for (int i = 0; i < 6; ++i) if (!num_arg(a[i], nm[i], r[i])) return nullptr;
		biquad_normalise(r[0], r[1], r[2], r[3], r[4], r[5], c);
		break;
	}
	}
	if (num != DSPAMD_BIQUAD_BIQUAD) biquad_design(type, is->fs, v[0], v[1], v[2], v[3], wt, c);
	if (*reverse) {
		// time-reversed IIR (biquad.c:532-533 -> reverse_iir.c:688-723): a section list per selected channel; consecutive
		// -r effects merge; prepare() turns the lists into one FIR per channel for the FFT convolver (reverse_iir.cpp)
		SpecPtr r = new_spec(Kind::Conv, name, is, sel);
		r->flags = EFFECT_FLAG_OPT_REORDERABLE | EFFECT_FLAG_CH_DEPS_IDENTITY;
		r->riir.assign(is->channels, std::vector<RiirSec>());
		r->riir_pending = true;
		RiirSec sec;
		riir_sec_from_biquad(c, thresh, &sec);
		for (int k = 0; k < is->channels; ++k) if (sel[k]) r->riir[k].push_back(sec);
		*reverse = false;                                   // provided
		return r;
	}
	SpecPtr s = new_spec(Kind::Biquad, name, is, sel);
	s->flags = EFFECT_FLAG_OPT_REORDERABLE | EFFECT_FLAG_CH_DEPS_IDENTITY;
	s->bq.assign(is->channels, std::array<double, 5>{ 1, 0, 0, 0, 0 });
	for (int k = 0; k < is->channels; ++k) if (sel[k]) s->bq[k] = c;
	return s;
}

// ------------------------------------------------------------- gain / add

SpecPtr parse_gain(int num, const stream_info *is, const char *sel, int argc, const char *const *argv)
{
	const char *name = argv[0];
	if (argc != 2) { usage(name); return nullptr; }
	char *end;
	double v = strtod(argv[1], &end);
	const char *what = (num == DSPAMD_GAIN_GAIN) ? "gain" : (num == DSPAMD_GAIN_MULT) ? "multiplier" : "value";
	if (bad_endptr(name, argv[1], end, what)) return nullptr;
	if (num == DSPAMD_GAIN_GAIN) v = pow(10.0, v / 20.0);   // gain.c:94
	const bool is_add = (num == DSPAMD_GAIN_ADD);
	SpecPtr s = new_spec(is_add ? Kind::Add : Kind::Gain, name, is, sel);
	s->flags = EFFECT_FLAG_CH_DEPS_IDENTITY | (is_add ? 0 : EFFECT_FLAG_OPT_REORDERABLE);
	s->vec.resize(is->channels);
	for (int k = 0; k < is->channels; ++k) s->vec[k] = sel[k] ? v : (is_add ? 0.0 : 1.0);
	return s;
}

// ------------------------------------------------------------------ remix

// remix.c:123-215: one selector per selected input channel position; extra selectors append channels;
// "." leaves an output silent; unselected channels pass through.
SpecPtr parse_remix(const stream_info *is, const char *sel, int argc, const char *const *argv)
{
	const char *name = argv[0];
	if (argc <= 1) { usage(name); return nullptr; }
	const int nin = is->channels;
	const Selector mask = copy_sel(sel, nin);
	const int n_sel = argc - 1, mask_bits = num_set(mask);
	const int nout = nin + n_sel - mask_bits;
	if (nout < 1) { set_error("%s: error: no output channels", name); return nullptr; }
	SpecPtr s = new_spec(Kind::Remix, name, is, sel);
	s->ch_out = nout;
	s->flags = EFFECT_FLAG_PLOT_MIX;
	s->remix.assign(nout, Selector(nin, 0));
	bool pure_routing = true;
	for (int k = 0, i = 0, ch = 0; k < nout; ++k, ++ch) {
		if (ch >= nin || mask[ch]) {
			if (i < n_sel) {
				if (strcmp(argv[i + 1], ".") != 0 && !parse_selector_masked(argv[i + 1], s->remix[k], mask, nin))
					return nullptr;
				if (num_set(s->remix[k]) > 1) pure_routing = false;
				++i;
			}
			else {
				while (ch < nin && mask[ch]) ++ch;
				if (ch < nin) s->remix[k][ch] = 1;
			}
		}
		else s->remix[k][ch] = 1;
	}
	if (pure_routing) s->flags |= EFFECT_FLAG_NO_DITHER;
	return s;
}

// ------------------------------------------------------------------ st2ms / ms2st / crossfeed

static bool pair_of(const char *name, const stream_info *is, const char *sel, int *c0, int *c1)
{
	*c0 = *c1 = -1;
	int n = 0;
	for (int k = 0; k < is->channels; ++k) if (sel[k]) { if (*c0 < 0) *c0 = k; else *c1 = k; ++n; }
	if (n != 2) { set_error("%s: error: parameter out of range: input channels must be 2", name); return false; }   // st2ms.c:88-91, crossfeed.c:99-102
	return true;
}

static void mix_identity(Spec &s, int n_out, int n_in)
{
	s.mix_idx.assign(n_out, std::vector<int>());
	s.mix_w.assign(n_out, std::vector<double>());
	s.remix.assign(n_out, Selector(n_in, 0));
	for (int k = 0; k < n_out && k < n_in; ++k) { s.mix_idx[k] = { k }; s.mix_w[k] = { 1.0 }; s.remix[k][k] = 1; }
}

static void mix_row(Spec &s, int k, std::vector<int> idx, std::vector<double> w)
{
	std::fill(s.remix[k].begin(), s.remix[k].end(), 0);
	for (int c : idx) s.remix[k][c] = 1;
	s.mix_idx[k] = std::move(idx);
	s.mix_w[k] = std::move(w);
}

// st2ms.c:28-54: (s0 + s1) * 0.5, (s0 - s1) * 0.5 on the selected pair; ms2st: s0 + s1, s0 - s1.  Bit-exact.
SpecPtr parse_st2ms(int num, const stream_info *is, const char *sel, int argc, const char *const *argv)
{
	const char *name = argv[0];
	if (argc != 1) { usage(name); return nullptr; }
	int c0, c1;
	if (!pair_of(name, is, sel, &c0, &c1)) return nullptr;
	if (num != DSPAMD_ST2MS_ST2MS && num != DSPAMD_ST2MS_MS2ST) { set_error("%s: error: effect number out of range", name); return nullptr; }
	SpecPtr s = new_spec(Kind::Mix, name, is, sel);
	s->flags = EFFECT_FLAG_PLOT_MIX;
	mix_identity(*s, is->channels, is->channels);
	mix_row(*s, c0, { c0, c1 }, { 1.0, 1.0 });
	mix_row(*s, c1, { c0, c1 }, { 1.0, -1.0 });
	if (num == DSPAMD_ST2MS_ST2MS) {
		s->mix_post.assign(is->channels, 1.0);
		s->mix_post[c0] = s->mix_post[c1] = 0.5;
	}
	return s;
}

// crossfeed.c:89-153
SpecPtr parse_crossfeed(const stream_info *is, const char *sel, int argc, const char *const *argv)
{
	const char *name = argv[0];
	if (argc != 3) { usage(name); return nullptr; }
	int c0, c1;
	if (!pair_of(name, is, sel, &c0, &c1)) return nullptr;
	char *end;
	const double freq = parse_freq(argv[1], &end);
	if (bad_endptr(name, argv[1], end, "f0")) return nullptr;
	if (!(freq >= 0.0 && freq < is->fs / 2.0)) { set_error("%s: error: parameter out of range: f0", name); return nullptr; }
	const double sep_db = strtod(argv[2], &end);
	if (bad_endptr(name, argv[2], end, "separation")) return nullptr;
	if (!(sep_db >= 0.0)) { set_error("%s: error: parameter out of range: separation", name); return nullptr; }
	SpecPtr s = new_spec(Kind::Crossfeed, name, is, sel);
	s->flags = EFFECT_FLAG_PLOT_MIX;
	s->xf_c0 = c0; s->xf_c1 = c1;
	const double sep = pow(10, sep_db / 20);
	s->xf_direct = sep / (1 + sep);
	s->xf_cross = 1 / (1 + sep);
	biquad_design(DSPAMD_BIQUAD_LOWPASS_1, is->fs, freq, 0, 0, 0, W_Q, s->xf_lp);
	biquad_design(DSPAMD_BIQUAD_HIGHPASS_1, is->fs, freq, 0, 0, 0, W_Q, s->xf_hp);
	s->remix.assign(is->channels, Selector(is->channels, 0));
	for (int k = 0; k < is->channels; ++k) s->remix[k][k] = 1;
	s->remix[c0][c1] = s->remix[c1][c0] = 1;                         // crossfeed.c:82-87
	return s;
}

// One crossfeed = three device stages over a stream widened by four scratch channels:
//   spread   C -> C+4   scratch = (s1, s0, s0, s1): the inputs of lp[0], hp[0], lp[1], hp[1] (crossfeed.c:41-46)
//   filters  the four first-order sections, fused into one cascade launch
//   combine  C+4 -> C   out_c0 = s0 direct + lp[0](s1) cross + hp[0](s0) cross, out_c1 likewise, in the reference's order
void crossfeed_expand(const Spec &xf, Spec &spread, Spec &filters, Spec &combine)
{
	const int C = xf.ch_in, W = C + 4, c0 = xf.xf_c0, c1 = xf.xf_c1;
	spread = Spec(); filters = Spec(); combine = Spec();
	spread.kind = Kind::Mix; spread.name = xf.name + ":spread";
	spread.fs_in = spread.fs_out = xf.fs_in; spread.ch_in = C; spread.ch_out = W;
	spread.sel = xf.sel;
	mix_identity(spread, W, C);
	const int src[4] = { c1, c0, c0, c1 };
	for (int j = 0; j < 4; ++j) mix_row(spread, C + j, { src[j] }, { 1.0 });
	filters.kind = Kind::Biquad; filters.name = xf.name + ":filters";
	filters.fs_in = filters.fs_out = xf.fs_in; filters.ch_in = filters.ch_out = W;
	filters.sel.assign(W, 0);
	filters.bq.assign(W, std::array<double, 5>{ { 1, 0, 0, 0, 0 } });
	for (int j = 0; j < 4; ++j) { filters.sel[C + j] = 1; filters.bq[C + j] = (j & 1) ? xf.xf_hp : xf.xf_lp; }
	combine.kind = Kind::Mix; combine.name = xf.name + ":combine";
	combine.fs_in = combine.fs_out = xf.fs_in; combine.ch_in = W; combine.ch_out = C;
	combine.sel.assign(W, 0);
	mix_identity(combine, C, W);
	mix_row(combine, c0, { c0, C + 0, C + 1 }, { xf.xf_direct, xf.xf_cross, xf.xf_cross });
	mix_row(combine, c1, { c1, C + 2, C + 3 }, { xf.xf_direct, xf.xf_cross, xf.xf_cross });
}

// ------------------------------------------------------------------ delay

SpecPtr make_delay_spec(const char *name, const stream_info *is, const char *sel, ssize_t samples, bool *noop)
{
	*noop = (samples == 0);   // delay.c:204-205: nothing to do -> run stays NULL
	SpecPtr s = new_spec(Kind::Delay, name, is, sel);
	s->flags = EFFECT_FLAG_OPT_REORDERABLE | EFFECT_FLAG_CH_DEPS_IDENTITY;
	s->delay.assign(is->channels, 0);
	s->delay_frac.assign(is->channels, 0.0);
	s->fd_ap_n.assign(is->channels, 0);
	for (int k = 0; k < is->channels; ++k) if (sel[k]) s->delay[k] = samples;
	return s;
}

// Thiran all-pass of order n and delay D (> n - 1) as second-order sections.  The reference runs the ladder of Koshita et
// al. (allpass.h:73-118, allpass.c:24-37); the transfer function is z^-n A(1/z) / A(z), so its poles are the roots of A: found in 113-bit
// arithmetic (thiran_roots.cpp), checked by multiplying the double-rounded factors back together, and paired into sections
// (c2 + c1 z^-1 + z^-2) / (1 + c1 z^-1 + c2 z^-2).  All of the reference's orders (1 .. 50, delay.c's option parser) pass the check; the outputs
// agree with the reference's ladder to 2e-14 RMS at n = 32 (scripts/exp_ap_orders.py, tests/test_gpu_parity.py).
static bool thiran_sections(int n, double D, std::vector<std::array<double, 5>> &out)
{
	double sec[33][5];
	const int got = dspamd_thiran_pole_sections(n, D, &sec[0][0], 33);
	if (got < 0) return false;
	out.clear();
	for (int i = 0; i < got; ++i) out.push_back({ sec[i][0], sec[i][1], sec[i][2], sec[i][3], sec[i][4] });
	return true;
}

// delay_effect_prepare (delay.c:149-204): what is left of the summed fractional amounts becomes a first- or
// second-order Thiran all-pass (allpass.h:46-71) -- the same transfer function as a biquad section
//     ap1: (c0 + z^-1) / (1 + c0 z^-1),   ap2: (c1 + c0 z^-1 + z^-2) / (1 + c0 z^-1 + c1 z^-2)
// which the fused cascade kernel runs; the integer remainder is requested from the host (channel_offsets).
bool delay_prepare(Spec &sp, bool *noop)
{
	*noop = true;
	if (sp.kind != Kind::Delay || sp.delay_frac.empty()) return true;
	const int n = sp.ch_in;
	bool any = false;
	for (int k = 0; k < n; ++k) {
		if (sp.fd_ap_n[k] < 1) sp.fd_ap_n[k] = 2;                       // DELAY_FD_AP_N_DEFAULT
		if (fabs(sp.delay_frac[k] - rint(sp.delay_frac[k])) >= DBL_EPSILON) {
			const ssize_t adj = (sp.fd_ap_n[k] - 1) - (ssize_t) floor(sp.delay_frac[k] - 0.1);   // DELAY_MIN_FRAC
			sp.delay[k] -= adj;
			sp.delay_frac[k] += adj;
			any = true;
		}
		else {
			sp.delay[k] += lrint(sp.delay_frac[k]);
			sp.delay_frac[k] = 0.0;
			sp.fd_ap_n[k] = 0;
		}
	}
	if (!any) return true;                                              // integer amounts only: run() stays a no-op
	sp.kind = Kind::Biquad;
	sp.frac_delay = true;
	sp.bq.assign(n, std::array<double, 5>{ 1, 0, 0, 0, 0 });
	sp.bq_more.clear(); sp.sel_more.clear();
	for (int k = 0; k < n; ++k) {
		sp.sel[k] = sp.fd_ap_n[k] > 0 ? 1 : 0;
		if (!sp.sel[k]) continue;
		const double d = fabs(sp.delay_frac[k]);
		if (sp.fd_ap_n[k] == 1) {                                       // allpass.h:46-56
			const double c0 = (1.0 - d) / (1.0 + d);
			sp.bq[k] = { c0, 1.0, 0.0, c0, 0.0 };
		}
		else if (sp.fd_ap_n[k] == 2) {                                  // allpass.h:58-71
			const double c0 = (4.0 - 2.0 * d) / (1.0 + d), c1 = ((d - 2.0) * (d - 1.0)) / ((d + 1.0) * (d + 2.0));
			sp.bq[k] = { c1, c0, 1.0, c0, c1 };
		}
		else {
			std::vector<std::array<double, 5>> secs;
			if (!thiran_sections(sp.fd_ap_n[k], d, secs)) {
				set_error("%s: error: all-pass order %d: the second-order factorisation of the Thiran filter did not reproduce its polynomial", sp.name.c_str(), sp.fd_ap_n[k]);
				return false;
			}
			sp.bq[k] = secs[0];
			for (size_t i = 1; i < secs.size(); ++i) {
				if (sp.bq_more.size() < i) { sp.bq_more.emplace_back(n, std::array<double, 5>{ 1, 0, 0, 0, 0 }); sp.sel_more.emplace_back(n, 0); }
				sp.bq_more[i - 1][k] = secs[i];
				sp.sel_more[i - 1][k] = 1;
			}
		}
	}
	*noop = false;
	return true;
}

// delay_effect_init_frac (delay.c:254-257): the whole amount is "fractional" until prepare()
SpecPtr make_frac_delay_spec(const char *name, const stream_info *is, const char *sel, double samples_frac, int fd_ap_n, bool *noop)
{
	SpecPtr s = make_delay_spec(name, is, sel, 0, noop);
	*noop = (samples_frac == 0.0);
	for (int k = 0; k < is->channels; ++k) if (sel[k]) { s->delay_frac[k] = samples_frac; s->fd_ap_n[k] = fd_ap_n; }
	return s;
}

SpecPtr parse_delay(const stream_info *is, const char *sel, int argc, const char *const *argv, bool *noop)
{
	const char *name = argv[0];
	GetOpt g;
	int opt;
	bool do_frac = false;
	int order = 0;
	while ((opt = g.next(argc - 1, argv, "f::m:M:b:q:")) != -1) {
		if (opt == 'f') {                                               // delay.c:696-703
			do_frac = true;
			if (g.arg) {
				char *e2;
				order = (int) strtol(g.arg, &e2, 10);
				if (bad_endptr(name, g.arg, e2, "order")) return nullptr;
				if (!(order > 0 && order <= 50)) { set_error("%s: error: parameter out of range: order", name); return nullptr; }
			}
			continue;
		}
		if (opt == 'm' || opt == 'M' || opt == 'b' || opt == 'q') {
			set_error("%s: error: option -%c (modulated delay, delay.c:567-593) is not provided by the GPU backend", name, opt);
			return nullptr;
		}
		g.print_error(opt, name);
		usage(name);
		return nullptr;
	}
	if (g.ind != argc - 1) { usage(name); return nullptr; }
	char *end;
	const double samples = parse_len_frac(argv[g.ind], is->fs, &end);
	if (bad_endptr(name, argv[g.ind], end, "delay")) return nullptr;
	if (do_frac) return make_frac_delay_spec(name, is, sel, samples, order, noop);
	const ssize_t si = (ssize_t) lrint(samples);
	if (fabs(samples - si) >= DBL_EPSILON)
		log_msg(LL_VERBOSE, "%s: info: delay rounded to %gs (%zd samples)", name, (double) si / is->fs, si);
	return make_delay_spec(name, is, sel, si, noop);
}

SpecPtr make_align_spec(int fs, int channels, const std::vector<ssize_t> &len, ssize_t discard)
{
	SpecPtr s(new Spec);
	s->kind = Kind::Align;
	s->name = "align";
	s->fs_in = s->fs_out = fs;
	s->ch_in = s->ch_out = channels;
	s->sel.assign(channels, 1);
	s->flags = EFFECT_FLAG_CH_DEPS_IDENTITY;
	s->delay = len;
	s->discard = discard;
	return s;
}

// -------------------------------------------------------------------- fir

SpecPtr make_fir_spec(const char *name, const stream_info *is, const char *sel, const double *filter, int fch, ssize_t T, ssize_t ref,
	int mode, int force_direct, int part_len)
{
	const int nsel = num_set(sel, is->channels);
	if (fch != 1 && fch != nsel) {
		set_error("%s: error: channels mismatch: channels=%d filter_channels=%d", name, nsel, fch);
		return nullptr;
	}
	if (T < 1) { set_error("%s: error: filter length must be >= 1", name); return nullptr; }
	// fir.c:241 (<= 16 taps), fir_p.c:364 (<= 32 taps forces the direct form)
	const bool direct = (mode != CONV_ZITA_EQUIV) && (force_direct || (mode == CONV_LATENCY_LEN && T <= 16) || (mode == CONV_ZERO_LATENCY && T <= 32));
	SpecPtr s = new_spec(direct ? Kind::FirDirect : Kind::Conv, name, is, sel);
	s->flags = EFFECT_FLAG_OPT_REORDERABLE | EFFECT_FLAG_CH_DEPS_IDENTITY;
	s->taps.assign(filter, filter + T * fch);
	s->fch = fch;
	s->T = T;
	s->ref = ref;
	s->conv_mode = direct ? CONV_ZERO_LATENCY : mode;
	if (!direct && mode == CONV_LATENCY_LEN) s->latency = next_fast_fftw_len(T);        // fir.c:303, 208-217
	if (mode == CONV_ZITA_EQUIV) s->latency = part_len > 0 ? part_len : 64;            // zita_convolver.cpp:93-102, README.md:428
	return s;
}

struct FirOpts {
	bool do_align = false;
	ssize_t offset = 0;
	const char *type = nullptr, *enc = nullptr;
	int channels = 0;
	bool big_endian = false, endian_given = false, endian_native = false;
	bool any_fs = false;             // `-r any` (fir_util.c:148-150: p.fs = 0); otherwise p.fs = the stream's rate (:130) and a container's own rate must match (:103-109)
};

// option grammar of fir_util.c:122-185 ("a::t:e:BLNr:c:")
static bool parse_fir_opts(const char *name, const stream_info *is, GetOpt &g, int argc, const char *const *argv, FirOpts &o)
{
	int opt;
	char *end;
	o.channels = is->channels;
	while ((opt = g.next(argc - 1, argv, "a::t:e:BLNr:c:")) != -1) {
		switch (opt) {
		case 'a':
			o.do_align = true;
			if (g.arg) {
				o.offset = parse_len(g.arg, is->fs, &end);
				if (bad_endptr(name, g.arg, end, "offset")) return false;
			}
			break;
		case 't': o.type = g.arg; break;
		case 'e': o.enc = g.arg; break;
		case 'B': o.big_endian = true; o.endian_given = true; o.endian_native = false; break;
		case 'L': o.big_endian = false; o.endian_given = true; o.endian_native = false; break;
		case 'N': o.big_endian = false; o.endian_given = true; o.endian_native = true; break;
		case 'r':
			if (strcmp(g.arg, "any") == 0) o.any_fs = true;
			else {
				const long fs = lround(parse_freq(g.arg, &end));
				if (bad_endptr(name, g.arg, end, "sample rate")) return false;
				if (fs <= 0) { set_error("%s: error: sample rate must be > 0", name); return false; }
				if (fs != is->fs) { set_error("%s: error: sample rate mismatch: stream_fs=%d requested_fs=%ld", name, is->fs, fs); return false; }
				o.any_fs = false;
			}
			break;
		case 'c':
			o.channels = (int) strtol(g.arg, &end, 10);
			if (bad_endptr(name, g.arg, end, "number of channels")) return false;
			if (o.channels <= 0) { set_error("%s: error: number of channels must be > 0", name); return false; }
			break;
		default:
			g.print_error(opt, name);
			return false;
		}
	}
	return true;
}

// RIFF/WAVE filter files (the usual export of room-correction tools).  The reference reads them through libsndfile
// (codec.c:83, sndfile.c: sf_readf_double with libsndfile's default normalisation), i.e. integer PCM divided by 2^(bits-1),
// unsigned 8-bit as (v - 128) / 128, float and double as stored; channel count and rate are the file's.  Formats: PCM
// 8/16/24/32 bit, IEEE float 32/64 bit, plain and WAVE_FORMAT_EXTENSIBLE headers, little-endian RIFF.
static bool read_wav(const char *name, const std::string &path, const std::vector<unsigned char> &raw, std::vector<double> &data, int *fch, ssize_t *T, int *fs)
{
	auto u16 = [&](size_t o) { return (unsigned) raw[o] | ((unsigned) raw[o + 1] << 8); };
	auto u32 = [&](size_t o) { return (uint32_t) raw[o] | ((uint32_t) raw[o + 1] << 8) | ((uint32_t) raw[o + 2] << 16) | ((uint32_t) raw[o + 3] << 24); };
	if (raw.size() < 12 || memcmp(&raw[0], "RIFF", 4) != 0 || memcmp(&raw[8], "WAVE", 4) != 0) {
		set_error("%s: error: not a RIFF/WAVE file: %s", name, path.c_str());
		return false;
	}
	unsigned tag = 0, channels = 0, bits = 0, align = 0;
	size_t d_off = 0, d_len = 0;
	bool have_fmt = false;
	for (size_t o = 12; o + 8 <= raw.size();) {
		const size_t len = u32(o + 4), body = o + 8;
		if (memcmp(&raw[o], "fmt ", 4) == 0 && len >= 16 && body + 16 <= raw.size()) {
			tag = u16(body); channels = u16(body + 2); *fs = (int) u32(body + 4); align = u16(body + 12); bits = u16(body + 14);
			if (tag == 0xFFFE && len >= 26 && body + 26 <= raw.size()) tag = u16(body + 24);    // extensible: first word of the sub-format GUID
			have_fmt = true;
		}
		else if (memcmp(&raw[o], "data", 4) == 0) {
			d_off = body;
			d_len = std::min(len, raw.size() - body);      // (streamed files carry 0xFFFFFFFF here)
			break;
		}
		o = body + len + (len & 1);
	}
	const unsigned bytes = bits / 8;
	if (!have_fmt || !d_off || channels < 1 || (tag != 1 && tag != 3) || bits % 8 || align != channels * bytes ||
	    (tag == 1 && (bytes < 1 || bytes > 4)) || (tag == 3 && bytes != 4 && bytes != 8)) {
		set_error("%s: error: unsupported WAVE format (PCM 8/16/24/32 bit and IEEE float 32/64 bit are read): %s", name, path.c_str());
		return false;
	}
	const size_t n = d_len / bytes / channels * channels;
	*fch = (int) channels;
	*T = (ssize_t) (n / channels);
	data.resize(n);
	for (size_t i = 0; i < n; ++i) {
		const unsigned char *b = &raw[d_off + i * bytes];
		if (tag == 3) {
			if (bytes == 4) { float v; memcpy(&v, b, 4); data[i] = (double) v; }
			else { double v; memcpy(&v, b, 8); data[i] = v; }
		}
		else if (bytes == 1) data[i] = ((int) b[0] - 128) / 128.0;
		else {
			int32_t v = 0;
			for (unsigned k = 0; k < bytes; ++k) v |= (int32_t) ((uint32_t) b[k] << (8 * (k + 4 - bytes)));   // left-justified in 32 bits
			data[i] = (double) v / 2147483648.0;
		}
	}
	return true;
}

static bool has_wav_ext(const std::string &path)
{
	const size_t dot = path.rfind('.');
	return dot != std::string::npos && (strcasecmp(path.c_str() + dot, ".wav") == 0 || strcasecmp(path.c_str() + dot, ".wavex") == 0);
}

// fir_util.c:25-120.  "coefs:a,b,c/d,e,f" literals, a raw PCM file (-t pcm -e double|float|s32|s24|s16, -c N), or a RIFF/WAVE
// file (-t wav | wavex | sndfile, or no -t and a .wav / .wavex name: codec.c:83, 200-211).  The other containers libsndfile /
// ffmpeg give the reference (flac, aiff, ...) are not read.
static bool read_filter(const char *name, const stream_info *is, const char *sel, const char *dir, const FirOpts &o, const char *spec,
	std::vector<double> &data, int *fch, ssize_t *T)
{
	if (strncmp(spec, "coefs:", 6) == 0) {
		const char *p = spec + 6;
		int channels = 1;
		ssize_t i = 1, frames = 1;
		for (const char *q = p; *q; ++q) {
			if (*q == ',') ++i;
			else if (*q == '/') { ++channels; if (i > frames) frames = i; i = 1; }
		}
		if (i > frames) frames = i;
		data.assign((size_t) frames * channels, 0.0);
		int ch = 0;
		std::string str(p);
		size_t pos = 0;
		while (pos <= str.size()) {
			size_t e = str.find('/', pos);
			if (e == std::string::npos) e = str.size();
			std::string chs = str.substr(pos, e - pos);
			size_t cp = 0;
			ssize_t idx = 0;
			while (cp <= chs.size()) {
				size_t ce = chs.find(',', cp);
				if (ce == std::string::npos) ce = chs.size();
				std::string tok = chs.substr(cp, ce - cp);
				const size_t a = tok.find_first_not_of(" \t"), b = tok.find_last_not_of(" \t");
				if (a != std::string::npos) {
					tok = tok.substr(a, b - a + 1);
					char *end;
					const double v = strtod(tok.c_str(), &end);
					if (bad_endptr(name, tok.c_str(), end, "coefficient")) return false;
					data[(size_t) idx * channels + ch] = v;
				}
				++idx;
				cp = ce + 1;
			}
			++ch;
			pos = e + 1;
			if (e == str.size()) break;
		}
		*fch = channels;
		*T = frames;
		return true;
	}
	const char *spec_as_given = spec;
	if (strncmp(spec, "file:", 5) == 0) spec += 5;
	std::string path = full_path(dir, spec, is->fs, num_set(copy_sel(sel, is->channels)));   // fir_util.c:85
	const bool wav = o.type ? (!strcmp(o.type, "wav") || !strcmp(o.type, "wavex") || !strcmp(o.type, "sndfile")) : has_wav_ext(path);
	const char *enc_own = o.enc ? o.enc : "s16";
	const bool pcm_own = o.type && !strcmp(o.type, "pcm") && (!strcmp(enc_own, "double") || !strcmp(enc_own, "float") || !strcmp(enc_own, "s32") || !strcmp(enc_own, "s24") || !strcmp(enc_own, "s16"));
	if (!wav && !pcm_own) {
		// not one of the formats decoded here: when this library sits in the reference host, the host's own fir_read_filter
		// (fir_util.c:25-120) reads the file through its codec layer -- whatever containers and encodings it was built with
		// (looked up per call, init time only: the host may come into scope after this library's first use in the process)
		const dspamd_host_fir_read_filter_fn host_read = reinterpret_cast<dspamd_host_fir_read_filter_fn>(dlsym(RTLD_DEFAULT, "fir_read_filter"));
		if (host_read) {
			effect_info ei;
			memset(&ei, 0, sizeof(ei));
			ei.name = name;
			dspamd_codec_params cp;
			memset(&cp, 0, sizeof(cp));
			cp.path = spec_as_given; cp.type = o.type; cp.enc = o.enc;
			cp.fs = o.any_fs ? 0 : is->fs;                       // fir_util.c:130, 148-150
			cp.channels = o.channels;
			cp.endian = o.endian_given ? (o.big_endian ? DSPAMD_CODEC_ENDIAN_BIG : o.endian_native ? DSPAMD_CODEC_ENDIAN_NATIVE : DSPAMD_CODEC_ENDIAN_LITTLE) : DSPAMD_CODEC_ENDIAN_DEFAULT;
			cp.mode = DSPAMD_CODEC_MODE_READ;
			cp.block_frames = 2048; cp.buf_ratio = 64;           // CODEC_PARAMS_AUTO, codec.h:62-68
			const std::vector<char> selc = copy_sel(sel, is->channels);
			int ch = 0;
			ssize_t fr = 0;
			sample_t *d = host_read(&ei, is, selc.data(), dir, &cp, &ch, &fr);
			if (!d) { set_error_quiet("%s: error: the host's fir_read_filter could not read: %s", name, spec); return false; }   // (the host has logged why: nothing more on stderr)
			data.assign(d, d + (size_t) fr * ch);
			free(d);
			*fch = ch; *T = fr;
			if (*T < 1) { set_error("%s: error: filter length must be >= 1", name); return false; }
			return true;
		}
		set_error("%s: error: filter files are read as raw PCM (-t pcm -e double|float|s32|s24|s16 -c N) or RIFF/WAVE (.wav) here; other formats need the reference host's codec layer: %s", name, spec);
		return false;
	}
	FILE *f = fopen(path.c_str(), "rb");
	if (!f) {
		// (the reference's codec layer names the container and the reason first: pcm.c / sndfile.c "failed to open file", then fir_util.c:115)
		log_msg(LL_ERROR, "%s: error: failed to open file: %s: %s", wav ? "sndfile" : "pcm", path.c_str(), strerror(errno));
		set_error("%s: error: failed to open filter file: %s", name, path.c_str());
		return false;
	}
	fseek(f, 0, SEEK_END);
	const long sz = ftell(f);
	fseek(f, 0, SEEK_SET);
	std::vector<unsigned char> raw(sz > 0 ? sz : 0);
	if (sz > 0 && fread(raw.data(), 1, raw.size(), f) != raw.size()) { fclose(f); set_error("%s: error: read failed: %s", name, path.c_str()); return false; }
	fclose(f);
	if (wav) {
		int file_fs = 0;
		if (!read_wav(name, path, raw, data, fch, T, &file_fs)) return false;
		if (file_fs != is->fs) {                      // fir_util.c:103-109: refused unless `-r any` was given
			if (!o.any_fs) {
				set_error("%s: error: sample rate mismatch: fs=%d filter_fs=%d", name, is->fs, file_fs);
				return false;
			}
			log_msg(4, "%s: info: ignoring sample rate mismatch: fs=%d filter_fs=%d", name, is->fs, file_fs);
		}
		if (*T < 1) { set_error("%s: error: filter length must be >= 1", name); return false; }
		return true;
	}
	const char *enc = o.enc ? o.enc : "s16";   // pcm.c:47 default
	int bytes;
	if (!strcmp(enc, "double")) bytes = 8;
	else if (!strcmp(enc, "float") || !strcmp(enc, "s32") || !strcmp(enc, "s24")) bytes = 4;
	else if (!strcmp(enc, "s16")) bytes = 2;
	else { set_error("%s: error: unsupported pcm encoding for filter files: %s", name, enc); return false; }
	const size_t n = raw.size() / bytes;
	*fch = o.channels;
	*T = (ssize_t) (n / o.channels);
	data.resize((size_t) *T * *fch);
	for (size_t i = 0; i < data.size(); ++i) {
		unsigned char b[8];
		memcpy(b, &raw[i * bytes], bytes);
		if (o.big_endian) for (int k = 0; k < bytes / 2; ++k) std::swap(b[k], b[bytes - 1 - k]);
		if (bytes == 8) { double v; memcpy(&v, b, 8); data[i] = v; }
		else if (!strcmp(enc, "float")) { float v; memcpy(&v, b, 4); data[i] = (double) v; }
		else if (!strcmp(enc, "s32")) { int32_t v; memcpy(&v, b, 4); data[i] = (double) v / 2147483648.0; }     // sampleconv.h:52
		else if (!strcmp(enc, "s24")) { int32_t v; memcpy(&v, b, 4); data[i] = (double) v / 8388608.0; }
		else { int16_t v; memcpy(&v, b, 2); data[i] = (double) v / 32768.0; }
	}
	if (*T < 1) { set_error("%s: error: filter length must be >= 1", name); return false; }
	(void) is; (void) sel;
	return true;
}

// fir_util.c:187-205
static ssize_t filter_offset(const FirOpts &o, const std::vector<double> &d, ssize_t T)
{
	if (!o.do_align) return 0;
	if (o.offset > 0) return o.offset;
	if (o.offset < 0) return T + o.offset;
	ssize_t off = 0;
	double peak = 0.0;
	for (size_t i = 0; i < d.size(); ++i) if (d[i] > peak) { peak = d[i]; off = (ssize_t) i; }
	return off;
}

static constexpr int FIR_P_DIRECT_LEN = 32;     // fir_p.c:34

SpecPtr parse_fir(const char *name, bool partitioned, const stream_info *is, const char *sel, const char *dir, int argc, const char *const *argv)
{
	GetOpt g;
	FirOpts o;
	if (!parse_fir_opts(name, is, g, argc, argv, o)) { usage(name); return nullptr; }
	long max_part_len = 0;
	if (partitioned) {
		if (g.ind < argc - 2 || g.ind > argc - 1) { usage(name); return nullptr; }
		if (g.ind == argc - 2) {
			char *end;
			max_part_len = strtol(argv[g.ind], &end, 10);
			if (bad_endptr(name, argv[g.ind], end, "max_part_len")) return nullptr;
			++g.ind;
		}
	}
	else if (g.ind != argc - 1) { usage(name); return nullptr; }
	std::vector<double> data;
	int fch;
	ssize_t T;
	if (!read_filter(name, is, sel, dir, o, argv[g.ind], data, &fch, &T)) return nullptr;
	const ssize_t ref = filter_offset(o, data, T);
	SpecPtr s = make_fir_spec(name, is, sel, data.data(), fch, T, ref, partitioned ? CONV_ZERO_LATENCY : CONV_LATENCY_LEN, 0, 0);
	if (s && partitioned && T > FIR_P_DIRECT_LEN && max_part_len != 0) {
		// fir_p.c:364-384: a filter of up to 32 taps takes the direct form before the value is looked at; otherwise powers of two in
		// [32, INT_MAX] (it only shapes the reference's latency-driven partition plan, which this engine does not need)
		if (max_part_len & (max_part_len - 1)) { set_error("%s: error: max_part_len must be a power of two", name); return nullptr; }
		if (max_part_len < FIR_P_DIRECT_LEN || max_part_len > 2147483647L) { set_error("%s: error: max_part_len must be within [%d,%d] or 0 for default", name, FIR_P_DIRECT_LEN, 2147483647); return nullptr; }
	}
	if (s) s->max_part_len = max_part_len;
	return s;
}

// How many filter terms the reference's fir_p carries: 32 direct taps, then up to four groups of equal partitions whose
// length grows by a step of 4 (8, 16 ... when four groups do not reach), each group covering the filter up to `step` (x2 when
// the group runs on a worker thread: filters of 4096 taps and more) times its partition length, the last group taking
// the rest; then groups trade partitions for doubled lengths where that saves transforms (fir_p.c:242-289).  The GPU engine
// plans its own transforms; this only fixes how many zero-padded terms `plot` prints.
ssize_t fir_p_planned_len(ssize_t T, long max_part_len)
{
	if (T <= 32) return T;
	const long cap = max_part_len ? max_part_len : (1L << 14);
	const long reach = (T < 4096) ? 1 : 2;
	struct Grp { long len, n; };
	std::vector<Grp> grp;
	for (long step = 4;; step *= 2) {
		grp.clear();
		bool fits = true;
		long len = 32;
		ssize_t covered = 32;
		while (covered < T) {
			if (grp.size() == 4) { fits = false; break; }
			Grp g = { len, 1 };
			covered += len;
			while (covered < T && covered < len * step * reach) { ++g.n; covered += len; }
			const long next = len * step;
			const bool last = next > cap || covered + next * step > T;
			if (last) while (covered < T) { ++g.n; covered += len; }
			grp.push_back(g);
			if (last) break;
			len = next;
		}
		if (fits) break;
	}
	for (size_t k = grp.size(); k-- > 1;) {
		Grp &g = grp[k], &before = grp[k - 1];
		while (g.len * 2 <= cap) {
			const long grown = before.n + g.len * reach / before.len;
			if (g.n <= grown) break;
			before.n = grown;
			g.len *= 2;
			g.n -= reach;
			g.n = g.n / 2 + (g.n & 1);
		}
	}
	ssize_t total = 32;
	for (const Grp &g : grp) total += (ssize_t) g.len * g.n;
	return total;
}

SpecPtr parse_zita(const stream_info *is, const char *sel, const char *dir, int argc, const char *const *argv)
{
	const char *name = argv[0];
	GetOpt g;
	FirOpts o;
	if (!parse_fir_opts(name, is, g, argc, argv, o) || g.ind < argc - 3 || g.ind > argc - 1) { usage(name); return nullptr; }
	long part[2] = { 0, 0 };
	char *end;
	for (int i = 0; g.ind < argc - 1 && i < 2; ++i, ++g.ind) {
		part[i] = strtol(argv[g.ind], &end, 10);
		if (bad_endptr(name, argv[g.ind], end, i ? "max_part_len" : "min_part_len")) return nullptr;
	}
	// Convproc::MINPART / MAXPART of zita-convolver 4.x are 64 / 8192
	const long minp = part[0] ? part[0] : 64, maxp = part[1] ? part[1] : 8192;
	if (minp < 64 || minp > 8192 || maxp < 64 || maxp > 8192 || (minp & (minp - 1))) {
		set_error("%s: error: partition lengths must be within [64,8192] or 0 for default", name);
		return nullptr;
	}
	std::vector<double> data;
	int fch;
	ssize_t T;
	if (!read_filter(name, is, sel, dir, o, argv[g.ind], data, &fch, &T)) return nullptr;
	const ssize_t ref = filter_offset(o, data, T);
	return make_fir_spec(name, is, sel, data.data(), fch, T, ref, CONV_ZITA_EQUIV, 0, (int) minp);
}

// ---------------------------------------------------------------- hilbert

void hilbert_design(ssize_t taps, double angle, std::vector<double> &h)
{
	// Blackman-windowed ideal Hilbert transformer mixed with a centre tap: hilbert.c:65-77
	h.assign(taps, 0.0);
	const double w_h = sin(-angle), w_d = cos(-angle);
	const ssize_t mid = taps / 2;
	for (ssize_t i = 0; i < taps; ++i) {
		const ssize_t k = i - mid;
		if (k == 0) h[i] = w_d;
		else if (k % 2 != 0) {
			const double x = 2.0*M_PI*i/(taps-1);
			h[i] = w_h * 2.0/(M_PI*k) * (0.42 - 0.5*cos(x) + 0.08*cos(2.0*x));
		}
	}
}

SpecPtr parse_hilbert(const stream_info *is, const char *sel, int argc, const char *const *argv)
{
	const char *name = argv[0];
	GetOpt g;
	int opt, conv = 0;
	bool do_align = false;
	double angle = -M_PI_2;
	char *end;
	while ((opt = g.next(argc - 1, argv, "pzca:")) != -1) {
		switch (opt) {
		case 'p': conv = 1; break;
		case 'z': conv = 2; break;
		case 'c': do_align = true; break;
		case 'a':
			angle = strtod(g.arg, &end) / 180.0 * M_PI;
			if (bad_endptr(name, g.arg, end, "angle")) return nullptr;
			break;
		default:
			g.print_error(opt, name);
			usage(name);
			return nullptr;
		}
	}
	if (g.ind != argc - 1) { usage(name); return nullptr; }
	const long taps = strtol(argv[g.ind], &end, 10);
	if (bad_endptr(name, argv[g.ind], end, "taps")) return nullptr;
	if (taps <= 3) { set_error("%s: error: taps must be > 3", name); return nullptr; }
	if (taps % 2 == 0) { set_error("%s: error: taps must be odd", name); return nullptr; }
	std::vector<double> h;
	hilbert_design(taps, angle, h);
	const ssize_t ref = do_align ? taps / 2 : 0;
	const int mode = (conv == 1) ? CONV_ZERO_LATENCY : (conv == 2) ? CONV_ZITA_EQUIV : CONV_LATENCY_LEN;
	return make_fir_spec(name, is, sel, h.data(), 1, taps, ref, mode, 0, 0);
}

// --------------------------------------------------------------- resample

static double rs_window(double x)
{
	// Albrecht 9-term window, resample.c:52-79 (WINDOW_FUNCTION 3)
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

static int gcd_i(int a, int b) { while (b) { const int t = b; b = a % b; a = t; } return a; }

SpecPtr parse_resample(const stream_info *is, const char *sel, int argc, const char *const *argv, bool *noop)
{
	const char *name = argv[0];
	*noop = false;
	if (argc < 2 || argc > 3) { usage(name); return nullptr; }
	const char *rate_arg = argv[argc - 1], *bw_arg = (argc == 3) ? argv[1] : nullptr;
	char *end;
	double bw = 0.939;
	if (bw_arg) {
		bw = strtod(bw_arg, &end);
		if (bad_endptr(name, bw_arg, end, "bandwidth")) return nullptr;
		if (!(bw >= 0.7 && bw <= 0.999)) { set_error("%s: error: parameter out of range: bandwidth", name); return nullptr; }
	}
	long rate;
	if (rate_arg[0] == 'x') {
		rate = is->fs * strtol(rate_arg + 1, &end, 10);
		if (bad_endptr(name, rate_arg, end, "fs multiplier")) return nullptr;
	}
	else if (rate_arg[0] == '/') {
		const long div = strtol(rate_arg + 1, &end, 10);
		if (bad_endptr(name, rate_arg, end, "fs divisor")) return nullptr;
		if (div == 0 || is->fs % div != 0) { set_error("%s: error: %ld is not a factor of %d", name, div, is->fs); return nullptr; }
		rate = is->fs / div;
	}
	else {
		rate = lround(parse_freq(rate_arg, &end));
		if (bad_endptr(name, rate_arg, end, "fs")) return nullptr;
	}
	if (rate <= 0) { set_error("%s: error: parameter out of range: rate", name); return nullptr; }
	SpecPtr s = new_spec(Kind::Resample, name, is, sel);
	s->sel.assign(is->channels, 1);            // resample ignores the selector (README.md:389-391)
	s->flags = EFFECT_FLAG_CH_DEPS_IDENTITY;
	s->fs_out = (int) rate;
	if (rate == is->fs) {
		log_msg(LL_VERBOSE, "%s: info: sample rates match; no proccessing will be done", name);
		*noop = true;
		return s;
	}
	// prototype design: resample.c:274-287, 363-364
	const double M_FACT = 17.7822;
	const int max_rate = rate > is->fs ? (int) rate : is->fs, min_rate = rate > is->fs ? is->fs : (int) rate;
	const int g = gcd_i((int) rate, is->fs);
	s->rs_n = (int) rate / g;
	s->rs_d = is->fs / g;
	const int min_factor = s->rs_n < s->rs_d ? s->rs_n : s->rs_d;
	const int m = (int) lround(2.0*M_FACT*max_rate / (min_rate*(1.0-bw)));
	const double width = M_FACT*max_rate / m;
	const double fc = (min_rate-width) / max_rate;
	const int os = min_factor < 2 ? min_factor : 2;
	const double fc_os = fc / os;
	const int m_os = (m + 1) * os - 1;
	s->rs_m = m;
	s->rs_fc = fc;
	s->rs_os = os;
	s->rs_proto.assign(m_os + 1, 0.0);
	for (int i = 1; i < m_os; ++i) {
		const double x = (i*2 - m_os)/2.0;
		const double sinc = (fabs(x) < 1e-9) ? fc_os : sin(M_PI*fc_os*x) / (M_PI*x);
		s->rs_proto[i] = sinc * rs_window((double) i / m_os);
	}
	log_msg(LL_VERBOSE, "%s: info: gcd=%d ratio=%d/%d width=%fHz fc=%f filter_len=%d sinc_oversample=%d", name, g, s->rs_n, s->rs_d, width, fc, m + 1, os);
	return s;
}

// ------------------------------------------------------------------ merge

// prepare() of a reverse-IIR effect (reverse_iir.c:381-636): per channel, the section list becomes one FIR
// (reverse_iir.cpp) and a delay the host compensates; channels without sections pass through.
bool riir_prepare(Spec &sp)
{
	if (!sp.riir_pending) return true;
	const int n = sp.ch_in;
	std::vector<std::vector<double>> h(n);
	sp.ch_latency.assign(n, 0);
	sp.riir_plot.assign(n, std::string());
	ssize_t T = 0;
	int nsel = 0;
	for (int k = 0; k < n; ++k) {
		sp.sel[k] = sp.riir[k].empty() ? 0 : 1;
		if (!sp.sel[k]) continue;
		if (!riir_design(sp.name.c_str(), k, sp.riir[k], h[k], &sp.ch_latency[k], &sp.riir_plot[k])) return false;
		T = std::max<ssize_t>(T, (ssize_t) h[k].size());
		++nsel;
	}
	sp.riir_pending = false;
	if (nsel == 0) { sp.T = 0; sp.fch = 1; sp.taps.clear(); return true; }
	// one filter per selected channel, in channel order (fir.c:342-357 mapping); a single shared filter when they agree
	bool same = true;
	int first = -1;
	for (int k = 0; k < n; ++k) {
		if (!sp.sel[k]) continue;
		if (first < 0) first = k;
		else if (h[k] != h[first]) same = false;
	}
	sp.T = T;
	sp.fch = same ? 1 : nsel;
	sp.taps.assign((size_t) T * sp.fch, 0.0);
	int f = 0;
	for (int k = 0; k < n; ++k) {
		if (!sp.sel[k]) continue;
		if (same && k != first) continue;
		for (size_t i = 0; i < h[k].size(); ++i) sp.taps[i * sp.fch + f] = h[k][i];
		++f;
	}
	sp.conv_mode = CONV_ZERO_LATENCY;
	sp.latency = 0;
	sp.ref = 0;
	return true;
}

bool merge_specs(Spec &d, const Spec &s)
{
	if (d.kind != s.kind) return false;
	const int n = d.ch_in;
	switch (d.kind) {
	case Kind::Gain:
		for (int k = 0; k < n; ++k) d.vec[k] *= s.vec[k];   // gain.c:57-67
		return true;
	case Kind::Add:
		for (int k = 0; k < n; ++k) d.vec[k] += s.vec[k];   // gain.c:69-79
		return true;
	case Kind::Biquad:
		for (int k = 0; k < n; ++k) if (d.sel[k] && s.sel[k]) return false;   // biquad.c:344-351
		for (int k = 0; k < n; ++k) if (s.sel[k]) { d.sel[k] = 1; d.bq[k] = s.bq[k]; }
		return true;
	case Kind::Delay:
		for (int k = 0; k < n; ++k) {                           // delay.c:127-141
			d.delay[k] += s.delay[k];
			d.delay_frac[k] += s.delay_frac[k];
			d.fd_ap_n[k] = std::max(d.fd_ap_n[k], s.fd_ap_n[k]);
		}
		return true;
	case Kind::Conv:
		// reverse IIR effects append their sections channel by channel (reverse_iir.c:305-319); nothing else merges
		if (!d.riir_pending || !s.riir_pending) return false;
		for (int k = 0; k < n; ++k) {
			d.riir[k].insert(d.riir[k].end(), s.riir[k].begin(), s.riir[k].end());
			if (!s.riir[k].empty()) d.sel[k] = 1;
		}
		return true;
	default:
		return false;
	}
}

}  // namespace dspamd
