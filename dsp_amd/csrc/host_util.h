// host_util.h -- logging, error reporting and the argument-parsing helpers the effects need.
// Behavioural mirrors (own code) of util.c:35-418: check_endptr, parse_freq, parse_len, selectors
// (one byte per channel, util.h:48-53) and the getopt clone with optional arguments ("x::").
#pragma once
#include <cstdarg>
#include <cstddef>
#include <string>
#include <vector>
#include <sys/types.h>

namespace dspamd {

enum { LL_SILENT = 0, LL_ERROR, LL_OPEN_ERROR, LL_NORMAL, LL_VERBOSE };  // dsp.h:25-32

extern int g_loglevel;
void log_msg(int level, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void set_error_quiet(const char *fmt, ...) __attribute__((format(printf, 1, 2)));   // last_error() only: the host has already said it on stderr
const char *last_error();

// returns true (and logs "failed to parse <what>: <str>") when the whole string was not consumed
bool bad_endptr(const char *name, const char *str, const char *endptr, const char *what);
double parse_freq(const char *s, char **endptr);                  // "1k" -> 1000
double parse_len_frac(const char *s, double fs, char **endptr);   // "10m" | "0.5s" | "37S" | plain seconds -> samples
ssize_t parse_len(const char *s, int fs, char **endptr);

using Selector = std::vector<char>;
bool parse_selector(const char *s, Selector &b, int n);                              // false on syntax error
bool parse_selector_masked(const char *s, Selector &b, const Selector &mask, int n);
int num_set(const Selector &b);
int num_set(const char *b, int n);

struct GetOpt {
	const char *arg = nullptr;
	int ind = 1, opt = 0, sp = 1;
	int next(int argc, const char *const *argv, const char *opts);   // -1 at end, '?' unknown, ':' missing argument
	void print_error(int r, const char *name) const;
};

ssize_t next_fast_fftw_len(ssize_t min_len);                      // util.c:434-458 (latency reported by `fir`)
std::string join_path(const char *dir, const char *path);         // "~/" and relative paths (util.c:276-343, no %-substitution)
std::string full_path(const char *dir, const char *path, int fs, int channels);   // + %r %k %c %% (util.c:276-343)
bool read_text_file(const std::string &path, std::string &out);
bool read_raw_doubles(const std::string &path, std::vector<double> &out);

}  // namespace dspamd
