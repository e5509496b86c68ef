// host_util.cpp -- see host_util.h
#include "host_util.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <dlfcn.h>

namespace dspamd {

static int initial_loglevel()
{
	const char *e = getenv("DSP_AMD_LOGLEVEL");   // a host's own -v / -q do not reach this library: LL_* number
	return e ? atoi(e) : LL_ERROR;
}

int g_loglevel = initial_loglevel();
static std::mutex g_log_mutex;
static thread_local char g_err[1024] = "";

// Linked into the reference host, messages carry ITS program name and obey ITS -v / -q: the host's `dsp_globals` (dsp.h:44-47,53:
// `struct dsp_globals { int loglevel; const char *prog_name; }`) is looked up in the process like `fir_read_filter` is (an executable
// exports it when linked with -rdynamic: INTEGRATION.md); the library's own level (DSP_AMD_LOGLEVEL, dspamd_set_loglevel) and name
// (DSP_AMD_PROG_NAME) serve a stand-alone host and override the host's when set.
struct HostGlobals { int loglevel; const char *prog_name; };
static const HostGlobals *host_globals()
{
	static const HostGlobals *g = static_cast<const HostGlobals *>(dlsym(RTLD_DEFAULT, "dsp_globals"));
	return g;
}

static const char *prog_name()
{
	static const char *env = getenv("DSP_AMD_PROG_NAME");
	if (env) return env;
	const HostGlobals *g = host_globals();
	return (g && g->prog_name) ? g->prog_name : "dsp";
}

static int effective_loglevel()
{
	static const bool own = getenv("DSP_AMD_LOGLEVEL") != nullptr;
	const HostGlobals *g = host_globals();
	return (g && !own && g_loglevel == LL_ERROR) ? g->loglevel : g_loglevel;      // (dspamd_set_loglevel moves g_loglevel off its default: the library's own then)
}

void log_msg(int level, const char *fmt, ...)
{
	if (effective_loglevel() < level) return;
	char buf[2048];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	std::lock_guard<std::mutex> lk(g_log_mutex);
	fprintf(stderr, "%s: %s\n", prog_name(), buf);
}

void set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	log_msg(LL_ERROR, "%s", g_err);
}

void set_error_quiet(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

const char *last_error() { return g_err; }

bool bad_endptr(const char *name, const char *str, const char *endptr, const char *what)
{
	if (endptr == str || *endptr != '\0') {
		set_error("%s%sfailed to parse %s: %s", name ? name : "", name ? ": " : "", what, str);
		return true;
	}
	return false;
}

double parse_freq(const char *s, char **r_endptr)
{
	char *end;
	double f = strtod(s, &end);
	if (end != s) {
		if (*end == 'k') {
			f *= 1000.0;
			++end;
		}
		if (*end != '\0') log_msg(LL_ERROR, "parse_freq: error: trailing characters: %s", end);     // util.c:58-59
	}
	if (r_endptr) *r_endptr = end;
	return f;
}

double parse_len_frac(const char *s, double fs, char **r_endptr)
{
	char *end;
	double d = strtod(s, &end);
	double samples = d * fs;
	if (end != s) {
		if (*end == 'm') { samples = d / 1000.0 * fs; ++end; }
		else if (*end == 's') { ++end; }
		else if (*end == 'S') { samples = d; ++end; }
		if (*end != '\0') log_msg(LL_ERROR, "parse_len_frac_2: error: trailing characters: %s", end);     // util.c:83-84
	}
	if (r_endptr) *r_endptr = end;
	return samples;
}

ssize_t parse_len(const char *s, int fs, char **r_endptr)
{
	return (ssize_t) lround(parse_len_frac(s, (double) fs, r_endptr));
}

// Channel selectors (README.md "Selector syntax"; verdicts as util.c:120-215): a comma-separated list of items
//   N | N-M | N- | -M | -        and the whole string "" or "-" = every channel.
// Parsed here item by item: split at the commas, then read `[number] ['-' [number]]` and demand the end of the item.
namespace {
struct SelItem { long lo = -1, hi = -1; bool range = false; };

// reads one item from [p, q); false + message on a syntax error
bool read_sel_item(const char *p, const char *q, SelItem &it)
{
	auto number = [&](long &out) {
		if (p == q || *p < '0' || *p > '9') return false;
		long v = 0;
		while (p != q && *p >= '0' && *p <= '9') { if (v < 100000000L) v = v * 10 + (*p - '0'); ++p; }
		out = v;
		return true;
	};
	if (p == q) { set_error("parse_selector: syntax error: ',' unexpected"); return false; }
	number(it.lo);
	if (p != q && *p == '-') {
		it.range = true;
		++p;
		number(it.hi);
	}
	if (p != q) {
		if (*p == '-') set_error("parse_selector: syntax error: '-' unexpected");
		else set_error("parse_selector: syntax error: invalid character: %c", *p);
		return false;
	}
	return true;
}
}  // namespace

bool parse_selector(const char *s, Selector &b, int n)
{
	if (strcmp(s, "") == 0 || strcmp(s, "-") == 0) { b.assign(n, 1); return true; }
	b.assign(n, 0);
	for (const char *item = s;;) {
		const char *stop = strchr(item, ',');
		const char *q = stop ? stop : item + strlen(item);
		SelItem it;
		if (!read_sel_item(item, q, it)) return false;
		for (long v : { it.lo, it.hi })
			if (v > n - 1) { set_error("parse_selector: error: value out of range: %ld", v); return false; }
		if (it.range && it.lo >= 0 && it.hi >= 0 && it.hi < it.lo) { set_error("parse_selector: error: malformed range: %ld-%ld", it.lo, it.hi); return false; }     // util.c:148
		// open ends: "-M" starts at 0, "N-" runs to the last channel, a lone number selects itself
		const long first = it.lo >= 0 ? it.lo : 0;
		const long last = it.hi >= 0 ? it.hi : (it.range ? n - 1 : it.lo);
		for (long c = first; c <= last; ++c) b[(size_t) c] = 1;
		if (!stop) break;
		item = stop + 1;
		if (*item == '\0') { set_error("parse_selector: syntax error: ',' unexpected"); return false; }
	}
	return true;
}

// a selector counted over the SET channels of `mask` (":0" inside a block that selected channels 2,3 means channel 2)
bool parse_selector_masked(const char *s, Selector &b, const Selector &mask, int n)
{
	std::vector<int> chosen;                       // chosen[i] = the i-th channel the mask lets through
	for (int c = 0; c < n && c < (int) mask.size(); ++c) if (mask[c]) chosen.push_back(c);
	Selector local;
	b.assign(n, 0);
	if (!parse_selector(s, local, (int) chosen.size())) return false;
	for (size_t i = 0; i < chosen.size(); ++i) if (local[i]) b[(size_t) chosen[i]] = 1;
	return true;
}

int num_set(const Selector &b)
{
	int c = 0;
	for (char v : b) if (v) ++c;
	return c;
}

int num_set(const char *b, int n)
{
	int c = 0;
	for (int i = 0; i < n; ++i) if (b[i]) ++c;
	return c;
}

// Option scanner with the reference's dialect (dsp_getopt, util.c): clustered flags ("-ab"), "x:" = required argument
// (attached or the next word), "x::" = optional argument (attached only), "--" ends the options; returns -1 at the
// first non-option word, '?' for a letter that is not in `opts`, ':' when a required argument is missing.
int GetOpt::next(int argc, const char *const *argv, const char *opts)
{
	enum Need { UNKNOWN, FLAG, REQUIRED, OPTIONAL };
	auto classify = [opts](int c) {
		if (c == ':' || c == '\0') return UNKNOWN;
		for (const char *o = opts; *o; ++o) {
			if (*o != c) continue;
			if (o[1] != ':') return FLAG;
			return o[2] == ':' ? OPTIONAL : REQUIRED;
		}
		return UNKNOWN;
	};
	if (sp == 1) {                                   // at the start of a word: is it an option word at all?
		const char *w = (ind < argc) ? argv[ind] : nullptr;
		if (!w || w[0] != '-' || w[1] == '\0') return -1;
		if (w[1] == '-' && w[2] == '\0') { ++ind; return -1; }
	}
	const char *word = argv[ind];
	const char *rest = word + sp + 1;                // what follows the letter inside this word
	opt = word[sp];
	auto step_inside_word = [&] { if (*rest) ++sp; else { ++ind; sp = 1; } };
	switch (classify(opt)) {
	case UNKNOWN:
		step_inside_word();
		return '?';
	case FLAG:
		arg = nullptr;
		step_inside_word();
		return opt;
	case OPTIONAL:
		arg = *rest ? rest : nullptr;
		++ind; sp = 1;
		return opt;
	case REQUIRED:
		sp = 1;
		if (*rest) { arg = rest; ++ind; return opt; }
		if (ind + 1 >= argc) { ++ind; return ':'; }
		arg = argv[ind + 1];
		ind += 2;
		return opt;
	}
	return -1;
}

void GetOpt::print_error(int r, const char *name) const
{
	set_error("%s: %s '%c'", name, (r == ':') ? "expected argument to option" : "unrecognized option", opt);
}

ssize_t next_fast_fftw_len(ssize_t min_len)
{
	ssize_t best = min_len * 7;
	const ssize_t bound = min_len * 2;
	for (ssize_t a = 1; a <= bound; a *= 2)
		for (ssize_t b = a; b <= bound; b *= 3)
			for (ssize_t c = b; c <= bound; c *= 5)
				for (ssize_t d = c; d <= bound; d *= 7)
					if (d >= min_len && d < best) best = d;
	return best;
}

std::string join_path(const char *dir, const char *path)
{
	std::string out;
	if (path[0] == '~' && path[1] == '/') {
		const char *home = getenv("HOME");
		if (home) out = home;
		out += (path + 1);
	}
	else if (dir && path[0] != '/') {
		out = dir;
		out += "/";
		out += path;
	}
	else out = path;
	return out;
}

// construct_full_path (util.c:276-343): `~/`, relative-to-dir, and the substitutions %r (rate), %k (rate / 1000),
// %c (channels), %% in effects-file and filter-file names
std::string full_path(const char *dir, const char *path, int fs, int channels)
{
	std::string sub;
	for (const char *q = path; *q; ++q) {
		if (q[0] == '%' && q[1] != '\0') {
			char buf[64];
			switch (q[1]) {
			case 'r': snprintf(buf, sizeof(buf), "%d", fs); sub += buf; ++q; continue;
			case 'k': snprintf(buf, sizeof(buf), "%.10g", fs / 1000.0); sub += buf; ++q; continue;
			case 'c': snprintf(buf, sizeof(buf), "%d", channels); sub += buf; ++q; continue;
			case '%': sub += '%'; ++q; continue;
			default: break;
			}
		}
		sub += *q;
	}
	return join_path(dir, sub.c_str());
}

bool read_text_file(const std::string &path, std::string &out)
{
	FILE *f = fopen(path.c_str(), "rb");
	if (!f) return false;
	char buf[4096];
	size_t n;
	out.clear();
	while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, n);
	fclose(f);
	return true;
}

bool read_raw_doubles(const std::string &path, std::vector<double> &out)
{
	FILE *f = fopen(path.c_str(), "rb");
	if (!f) return false;
	fseek(f, 0, SEEK_END);
	const long sz = ftell(f);
	fseek(f, 0, SEEK_SET);
	out.resize(sz > 0 ? (size_t) sz / sizeof(double) : 0);
	const size_t n = out.empty() ? 0 : fread(out.data(), sizeof(double), out.size(), f);
	fclose(f);
	return n == out.size();
}

}  // namespace dspamd
