// host_util.cpp -- see host_util.h
#include "host_util.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace dspamd {

static int initial_loglevel()
{
	const char *e = getenv("DSP_AMD_LOGLEVEL");   // a host's own -v / -q do not reach this library: LL_* number
	return e ? atoi(e) : LL_ERROR;
}

int g_loglevel = initial_loglevel();
static std::mutex g_log_mutex;
static thread_local char g_err[1024] = "";

static const char *prog_name()
{
	static const char *p = nullptr;
	if (!p) {
		p = getenv("DSP_AMD_PROG_NAME");
		if (!p) p = "dsp";
	}
	return p;
}

void log_msg(int level, const char *fmt, ...)
{
	if (g_loglevel < level) return;
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
	if (end != s && *end == 'k') {
		f *= 1000.0;
		++end;
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
	}
	if (r_endptr) *r_endptr = end;
	return samples;
}

ssize_t parse_len(const char *s, int fs, char **r_endptr)
{
	return (ssize_t) lround(parse_len_frac(s, (double) fs, r_endptr));
}

static void fill_range(Selector &b, int n, int start, int end, bool dash)
{
	if (start == -1 && end == -1) { start = 0; end = n - 1; }
	else if (start == -1) start = 0;
	else if (end == -1) end = dash ? n - 1 : start;
	for (int i = start; i <= end; ++i) b[i] = 1;
}

bool parse_selector(const char *s, Selector &b, int n)
{
	b.assign(n, 0);
	if (s[0] == '\0' || (s[0] == '-' && s[1] == '\0')) {
		b.assign(n, 1);
		return true;
	}
	int start = -1, end = -1;
	bool dash = false;
	for (;;) {
		if (*s >= '0' && *s <= '9') {
			const int v = atoi(s);
			if (v < 0 || v > n - 1) { set_error("parse_selector: error: value out of range: %d", v); return false; }
			if (dash) {
				if (v < start) { set_error("parse_selector: error: malformed range"); return false; }
				end = v;
			}
			else start = v;
			while (*s >= '0' && *s <= '9') ++s;
		}
		else if (*s == '-') {
			if (dash) { set_error("parse_selector: syntax error: '-' unexpected"); return false; }
			dash = true;
			++s;
		}
		else if (*s == ',' || *s == '\0') {
			if (start == -1 && end == -1 && !dash) { set_error("parse_selector: syntax error: ',' unexpected"); return false; }
			fill_range(b, n, start, end, dash);
			start = end = -1;
			dash = false;
			if (*s == '\0') break;
			++s;
			if (*s == '\0') { set_error("parse_selector: syntax error: ',' unexpected"); return false; }
		}
		else { set_error("parse_selector: syntax error: invalid character: %c", *s); return false; }
	}
	return true;
}

bool parse_selector_masked(const char *s, Selector &b, const Selector &mask, int n)
{
	b.assign(n, 0);
	const int nb = num_set(mask);
	Selector tmp;
	if (!parse_selector(s, tmp, nb)) return false;
	for (int i = 0, k = 0; i < nb; ++i, ++k) {
		while (k < n && !mask[k]) ++k;
		if (k == n) { set_error("parse_selector_masked(): BUG: too many channels"); return false; }   // util.c:203-207
		if (tmp[i]) b[k] = 1;
	}
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

int GetOpt::next(int argc, const char *const *argv, const char *opts)
{
	if (sp == 1) {
		if (ind >= argc || argv[ind][0] != '-' || argv[ind][1] == '\0') return -1;
		if (strcmp(argv[ind], "--") == 0) { ++ind; return -1; }
	}
	const int c = opt = argv[ind][sp];
	const char *cp = (c == ':') ? nullptr : strchr(opts, c);
	if (!cp) {
		if (argv[ind][++sp] == '\0') { ++ind; sp = 1; }
		return '?';
	}
	if (cp[1] == ':') {
		if (argv[ind][sp + 1] != '\0') arg = &argv[ind++][sp + 1];
		else if (cp[2] == ':') { ++ind; arg = nullptr; }           // optional argument, absent
		else if (++ind >= argc) { sp = 1; return ':'; }
		else arg = argv[ind++];
		sp = 1;
	}
	else {
		if (argv[ind][++sp] == '\0') { ++ind; sp = 1; }
		arg = nullptr;
	}
	return c;
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
