// chain.cpp -- see chain.h.  Behavioural mirror of the reference's chain runtime for hosts that do not link
// effects_chain.c:
//   chain language (effects, ":selector", "{ }" blocks, "!")       effects_chain.c:36-603, README.md:602-645
//   merge optimisation                                              effects_chain.c:605-641
//   per-channel latency alignment (auto-inserted align effects)     effects_chain.c:727-875, align.c:95-162
//   drain accounting                                                effects_chain.c:877-923
// It talks to the effects only through the plugin ABI (ei->init, e->merge, e->channel_offsets, ...), i.e. the
// same calls the reference host makes.
#include "chain.h"
#include <algorithm>
#include <cerrno>
#include <cstring>
#include <numeric>
#include <functional>

namespace dspamd {

ChainPlan::~ChainPlan()
{
	for (struct effect *e : effects) {
		if (e->destroy) e->destroy(e);
		free(e);
	}
}

std::vector<const Spec *> ChainPlan::specs() const
{
	std::vector<const Spec *> v;
	for (struct effect *e : effects) v.push_back(node_of(e)->spec.get());
	return v;
}

// ------------------------------------------------------------------ lexer

enum TokId { T_LITERAL, T_ESC_LITERAL, T_CH_SEL, T_BLOCK_START, T_BLOCK_END, T_SOURCE, T_ALLOW_FAIL };

struct Token {
	TokId id;
	std::string str;
};

static TokId classify(const std::string &s)
{
	if (!s.empty() && s[0] == ':') return T_CH_SEL;
	if (s == "{") return T_BLOCK_START;
	if (s == "}") return T_BLOCK_END;
	if (s.size() > 1 && s[0] == '@') return T_SOURCE;
	if (s == "!") return T_ALLOW_FAIL;
	return T_LITERAL;
}

static std::vector<Token> lex(const char *s)
{
	std::vector<Token> toks;
	while (*s) {
		while (*s == ' ' || *s == '\t' || *s == '\n' || *s == '\r') ++s;
		if (!*s) break;
		if (*s == '#') { while (*s && *s != '\n') ++s; continue; }
		std::string w;
		bool escaped_first = false, first = true;
		char quote = 0;
		while (*s) {
			if (quote) {
				if (*s == quote) { quote = 0; ++s; continue; }
				if (*s == '\\' && quote == '"' && s[1]) { w += s[1]; s += 2; continue; }
				w += *s++;
				continue;
			}
			if (*s == ' ' || *s == '\t' || *s == '\n' || *s == '\r') break;
			if (*s == '\\' && s[1]) { if (first) escaped_first = true; w += s[1]; s += 2; first = false; continue; }
			if (*s == '"' || *s == '\'') { quote = *s++; first = false; continue; }
			w += *s++;
			first = false;
		}
		Token t;
		t.id = escaped_first ? T_ESC_LITERAL : classify(w);
		t.str = (t.id == T_CH_SEL || t.id == T_SOURCE) ? w.substr(1) : w;
		toks.push_back(t);
	}
	return toks;
}

// names the reference's registry knows (effect.c:46-67) but this library does not provide
static const char *const k_unprovided[] = {
	"matrix4", "matrix4_mb", "decorrelate", "noise", "dither", "ladspa_host",
	"stats", "watch", "levels", nullptr,
};

static bool is_keyword(const Token &t)
{
	if (t.id == T_ESC_LITERAL) return false;
	if (t.id != T_LITERAL) return true;
	if (registry_lookup(t.str.c_str())) return true;
	for (int i = 0; k_unprovided[i]; ++i) if (t.str == k_unprovided[i]) return true;
	return false;
}

// ----------------------------------------------------------------- parser

struct Parser {
	ChainPlan *plan;
	stream_info *stream;
	const char *dir;
	const std::vector<Token> *toks;
};

// parses tokens[pos..) until end (or the matching '}' when in_block); returns index after the consumed
// tokens, or npos on error.  ch_mask: the channels this (sub-)chain may touch.
static size_t parse_tokens(Parser &P, size_t pos, Selector ch_mask, bool in_block, int depth)
{
	const std::vector<Token> &T = *P.toks;
	const size_t npos = (size_t) -1;
	if (depth > 32) { set_error("chain: error: maximum recursion depth exceeded"); return npos; }
	int last_ch = P.stream->channels;
	Selector ch_sel = ch_mask;
	std::string last_sel, ch_changed_by;      // ch_changed_by: the effect that last changed the channel count (prev_effect, effects_chain.c:504)
	bool have_last_sel = false, allow_fail = false;
	while (pos < T.size()) {
		const Token &tok = T[pos];
		if (in_block && tok.id == T_BLOCK_END) return pos;
		if (tok.id == T_ALLOW_FAIL) { allow_fail = true; ++pos; continue; }
		const int ch = P.stream->channels;
		if (last_ch != ch) {   // the channel count changed: rebuild the mask (effects_chain.c:460-488)
			Selector m(ch, 0);
			const int delta = ch - last_ch;
			if (delta > 0) {
				std::copy(ch_mask.begin(), ch_mask.end(), m.begin());
				for (int j = last_ch; j < ch; ++j) m[j] = 1;
			}
			else {
				int nb = num_set(ch_mask) + delta;
				for (int j = 0; j < ch && nb > 0; ++j) if (ch_mask[j]) { m[j] = 1; --nb; }
			}
			ch_mask = m;
		}
		if (tok.id == T_CH_SEL) {
			if (!parse_selector_masked(tok.str.c_str(), ch_sel, ch_mask, ch)) return npos;
			last_ch = ch;
			last_sel = tok.str;
			have_last_sel = true;
			++pos;
			continue;
		}
		if (last_ch != ch) {   // re-parse the active selector against the new channel count (:497-511)
			if (!have_last_sel) ch_sel = ch_mask;
			else if (!parse_selector_masked(last_sel.c_str(), ch_sel, ch_mask, ch)) {
				// the reference's two notes (effects_chain.c:503-504), folded into the message the C API hands back
				const std::string why = last_error();
				set_error("%s; note: active channel selector defined here: %s; note: number of channels modified by this effect: %s",
				          why.c_str(), last_sel.c_str(), ch_changed_by.empty() ? "?" : ch_changed_by.c_str());
				return npos;
			}
			last_ch = ch;
		}
		if (tok.id == T_SOURCE) {
			// ec_parse_file (effects_chain.c:336-372): the file's effects act on the active selection; relative paths
			// inside it are relative to ITS directory
			const std::string path = full_path(P.dir, tok.str.c_str(), P.stream->fs, num_set(ch_sel));
			std::string text;
			if (!read_text_file(path, text)) { set_error("error: failed to load effects file: %s: %s", path.c_str(), strerror(errno)); return npos; }
			const size_t slash = path.rfind('/');
			const std::string sub_dir = (slash == std::string::npos) ? "." : path.substr(0, slash);
			log_msg(LL_VERBOSE, "info: begin effects file: %s", path.c_str());
			const std::vector<Token> sub_toks = lex(text.c_str());
			Parser sub{ P.plan, P.stream, sub_dir.c_str(), &sub_toks };
			const size_t end = parse_tokens(sub, 0, ch_sel, false, depth + 1);
			if (end == npos) return npos;
			if (end != sub_toks.size()) { set_error("chain: error: unexpected token in %s: %s", path.c_str(), sub_toks[end].str.c_str()); return npos; }
			log_msg(LL_VERBOSE, "info: end effects file: %s", path.c_str());
			++pos;
			continue;
		}
		if (tok.id == T_BLOCK_START) {
			const size_t end = parse_tokens(P, pos + 1, ch_sel, true, depth + 1);
			if (end == npos) return npos;
			if (end >= T.size() || T[end].id != T_BLOCK_END) { set_error("chain: error: unterminated block"); return npos; }
			pos = end + 1;
			continue;
		}
		if (tok.id != T_LITERAL && tok.id != T_ESC_LITERAL) { set_error("chain: error: unexpected token: %s", tok.str.c_str()); return npos; }
		size_t end = pos + 1;
		while (end < T.size() && !is_keyword(T[end])) ++end;
		const effect_info *ei = registry_lookup(tok.str.c_str());
		if (!ei) {
			bool known = false;
			for (int i = 0; k_unprovided[i]; ++i) if (tok.str == k_unprovided[i]) known = true;
			log_msg(LL_ERROR, "%s: %s: %s", allow_fail ? "warning" : "error", known ? "effect not available" : "no such effect", tok.str.c_str());
			if (!allow_fail) { set_error("%s: %s", known ? "effect not available (not provided by the GPU backend)" : "no such effect", tok.str.c_str()); return npos; }
		}
		else {
			std::vector<const char *> argv;
			for (size_t i = pos; i < end; ++i) argv.push_back(T[i].str.c_str());
			struct effect *e = ei->init(ei, P.stream, ch_sel.data(), P.dir, (int) argv.size(), argv.data());
			if (!e) {
				log_msg(LL_ERROR, "%s: failed to initialize effect: %s", allow_fail ? "warning" : "error", tok.str.c_str());
				if (!allow_fail) return npos;
			}
			while (e) {   // an init may return a ->next linked list of sub-effects (effects_chain.c:584-596)
				struct effect *nx = e->next;
				e->next = nullptr;
				if (!e->run) {
					log_msg(LL_VERBOSE, "info: not using effect: %s", e->name ? e->name : tok.str.c_str());
					if (e->destroy) e->destroy(e);
					free(e);
				}
				else {
					P.plan->effects.push_back(e);
					if (e->ostream.channels != P.stream->channels) ch_changed_by = tok.str;
					*P.stream = e->ostream;
				}
				e = nx;
			}
		}
		allow_fail = false;
		pos = end;
	}
	if (in_block) return pos;   // caller reports the missing '}'
	return pos;
}

static void relink(ChainPlan &plan)
{
	for (size_t i = 0; i < plan.effects.size(); ++i) {
		plan.effects[i]->prev = i ? plan.effects[i - 1] : nullptr;
		plan.effects[i]->next = (i + 1 < plan.effects.size()) ? plan.effects[i + 1] : nullptr;
	}
}

// ----------------------------------------------------- optimise (merge)

static void optimize(ChainPlan &plan)
{
	auto &E = plan.effects;
	const size_t before = E.size();
	for (size_t d = 0; d < E.size(); ++d) {
		struct effect *dest = E[d];
		if (!dest->merge) continue;
		size_t s = d + 1;
		while (s < E.size()) {
			struct effect *src = E[s];
			if (src->istream.fs != dest->istream.fs || src->istream.channels != dest->istream.channels
					|| src->ostream.fs != dest->ostream.fs || src->ostream.channels != dest->ostream.channels)
				break;
			if (!src->merge) {
				if (src->flags & EFFECT_FLAG_OPT_REORDERABLE) { ++s; continue; }
				break;
			}
			if (dest->merge(dest, src)) {
				if (src->destroy) src->destroy(src);
				free(src);
				E.erase(E.begin() + s);
			}
			else ++s;   // a refused merge is skipped and the scan goes on (effects_chain.c:630-633)
		}
	}
	if (E.size() < before) log_msg(LL_VERBOSE, "optimize: info: reduced number of effects from %zu to %zu", before, E.size());
}

// ------------------------------------------------------------- alignment

struct DepMap {
	int n_in = 0, n_out = 0;
	std::vector<Selector> deps;   // [out] over in
};

static bool query_deps(struct effect *e, DepMap &m, int max_in, int max_out)
{
	if (!e->channel_deps) return false;
	m.n_in = e->istream.channels;
	m.n_out = e->ostream.channels;
	std::vector<std::vector<char>> store(max_out, std::vector<char>(max_in, 0));
	const int mn = std::min(m.n_in, m.n_out);
	for (int i = 0; i < mn; ++i) store[i][i] = 1;   // identity is the initial state
	std::vector<char *> ptrs(max_out);
	for (int i = 0; i < max_out; ++i) ptrs[i] = store[i].data();
	e->channel_deps(e, ptrs.data());
	m.deps.assign(m.n_out, Selector());
	for (int i = 0; i < m.n_out; ++i) m.deps[i].assign(store[i].begin(), store[i].begin() + m.n_in);
	return true;
}

// insert an align effect after effects[after] (align.c:95-162); offsets are updated in place
static bool insert_align(ChainPlan &plan, size_t after, std::vector<ssize_t> &offsets, const std::vector<ssize_t> *refs)
{
	struct effect *prev = plan.effects[after];
	const int n = prev->ostream.channels;
	const bool at_end = (after + 1 == plan.effects.size());
	const char *next_name = at_end ? "[end of chain]" : plan.effects[after + 1]->name;
	bool need = false;
	for (int k = 0; k < n; ++k)
		if (offsets[k] != (refs ? (*refs)[k] : 0)) { need = true; break; }
	if (!need) {
		log_msg(LL_VERBOSE, "info: no alignment needed: %s", next_name);
		return true;
	}
	ssize_t max_offset = at_end ? 0 : offsets[0];   // negative offsets are zeroed at the end of the chain
	for (int k = 0; k < n; ++k) max_offset = std::max(max_offset, offsets[k]);
	ssize_t min_ref = max_offset;
	std::vector<ssize_t> len(n, 0);
	for (int k = 0; k < n; ++k) {
		const ssize_t ref = refs ? (*refs)[k] : max_offset;
		min_ref = std::min(min_ref, ref);
		if (offsets[k] != ref) {
			len[k] = ref - offsets[k];
			log_msg(LL_VERBOSE, "align (%s): info: channel %d: %zd", next_name, k, len[k]);
		}
		offsets[k] = ref;
	}
	ssize_t discard = 0;
	if (min_ref > 0) {
		for (int k = 0; k < n; ++k) offsets[k] -= min_ref;
		discard = min_ref;
		log_msg(LL_VERBOSE, "align (%s): info: discarding %zd frames", next_name, discard);
	}
	struct effect *a = make_align_effect(prev->ostream.fs, n, len, discard);
	if (!a) return false;
	plan.effects.insert(plan.effects.begin() + after + 1, a);
	return true;
}

static int gcd_i(int a, int b) { while (b) { const int t = b; b = a % b; a = t; } return a; }
static ssize_t mult_ceil(ssize_t v, int n, int d) { const long long r = (long long) v * n; return (ssize_t) ((r % d) ? r / d + 1 : r / d); }

static bool align_channels(ChainPlan &plan)
{
	int max_in = 0, max_out = 0;
	for (struct effect *e : plan.effects) {
		max_in = std::max(max_in, e->istream.channels);
		max_out = std::max(max_out, e->ostream.channels);
	}
	const int max_ch = std::max(max_in, max_out);
	std::vector<ssize_t> offsets(max_ch, 0), delays(max_ch, 0);
	ssize_t nd_part = 0;   // negative part of the requested delays
	for (size_t i = 0; i < plan.effects.size(); ++i) {
		struct effect *e = plan.effects[i];
		const int nin = e->istream.channels, nout = e->ostream.channels;
		const bool passthrough = (nin == nout) && (e->flags & (EFFECT_FLAG_CH_DEPS_IDENTITY | EFFECT_FLAG_OPT_REORDERABLE));
		DepMap dm;
		const bool have_deps = query_deps(e, dm, max_in, max_out);
		if (i > 0) {
			const size_t n_before = plan.effects.size();
			if (e->flags & EFFECT_FLAG_ALIGN_BARRIER) {
				if (!insert_align(plan, i - 1, offsets, nullptr)) return false;
			}
			else if (have_deps) {
				// channels that end up mixed together (transitively) must share one offset: connected components
				std::vector<int> comp(nin);
				std::iota(comp.begin(), comp.end(), 0);
				std::function<int(int)> find = [&](int x) { while (comp[x] != x) x = comp[x] = comp[comp[x]]; return x; };
				for (int o = 0; o < nout; ++o) {
					int first = -1;
					for (int k = 0; k < nin; ++k) if (dm.deps[o][k]) { if (first < 0) first = k; else comp[find(k)] = find(first); }
				}
				std::vector<ssize_t> refs(offsets.begin(), offsets.begin() + max_ch);
				std::vector<ssize_t> cmax(nin);
				for (int k = 0; k < nin; ++k) cmax[k] = offsets[k];
				for (int k = 0; k < nin; ++k) { const int r = find(k); cmax[r] = std::max(cmax[r], offsets[k]); }
				for (int k = 0; k < nin; ++k) refs[k] = cmax[find(k)];
				if (!insert_align(plan, i - 1, offsets, &refs)) return false;
			}
			else if (e->istream.fs != e->ostream.fs) {
				log_msg(LL_VERBOSE, "info: %s: sample rate changed; doing full alignment", e->name);
				if (!insert_align(plan, i - 1, offsets, nullptr)) return false;
			}
			else if (!passthrough) {
				log_msg(LL_VERBOSE, "warning: %s: channel deps unknown; doing full alignment", e->name);
				if (!insert_align(plan, i - 1, offsets, nullptr)) return false;
			}
			if (plan.effects.size() != n_before) ++i;   // e moved one slot to the right
		}
		if (have_deps) {
			const std::vector<ssize_t> t_off(offsets), t_del(delays);
			ssize_t max_offset = 0;
			for (int k = 0; k < nin; ++k) max_offset = std::max(max_offset, t_off[k]);
			for (int o = 0; o < nout; ++o) {
				int first = -1;
				delays[o] = 0;
				for (int k = 0; k < nin; ++k) {
					if (!dm.deps[o][k]) continue;
					if (first < 0) { first = k; delays[o] = t_del[k]; }
					else if (t_off[k] != t_off[first]) { set_error("align: BUG: channel %d offset incorrect", k); return false; }
					else delays[o] = std::min(delays[o], t_del[k]);
				}
				offsets[o] = (first >= 0) ? t_off[first] : max_offset;
			}
		}
		else if (!passthrough) {
			ssize_t min_delay = delays[0];
			for (int k = 1; k < nin; ++k) {
				min_delay = std::min(min_delay, delays[k]);
				if (offsets[k] != offsets[k - 1]) { set_error("align: BUG: channel %d offset incorrect", k); return false; }
			}
			for (int o = 0; o < nout; ++o) delays[o] = min_delay;
		}
		for (int k = nout; k < nin; ++k) delays[k] = offsets[k] = 0;
		for (int o = 0; o < nout; ++o) offsets[o] += delays[o] - nd_part;   // cumulative latency
		if (e->channel_offsets) e->channel_offsets(e, offsets.data(), delays.data());
		else if (e->ostream.fs != e->istream.fs) {
			const int g = gcd_i(e->ostream.fs, e->istream.fs);
			for (int o = 0; o < nout; ++o) delays[o] = mult_ceil(delays[o], e->ostream.fs / g, e->istream.fs / g);
		}
		nd_part = 0;
		for (int o = 0; o < nout; ++o) nd_part = std::min(nd_part, delays[o]);
		for (int o = 0; o < nout; ++o) offsets[o] -= delays[o] - nd_part;
	}
	plan.zero_ref = -nd_part;
	if (!plan.effects.empty() && !insert_align(plan, plan.effects.size() - 1, offsets, nullptr)) return false;
	return true;
}

// ----------------------------------------------------------------- drain

static void compute_drain(ChainPlan &plan)
{
	plan.drain_frames = 0;
	if (plan.effects.empty()) return;
	int max_in = 0, max_out = 0;
	for (struct effect *e : plan.effects) {
		max_in = std::max(max_in, e->istream.channels);
		max_out = std::max(max_out, e->ostream.channels);
	}
	const int max_ch = std::max(max_in, max_out);
	std::vector<ssize_t> samples(max_ch, 0);
	for (struct effect *e : plan.effects) {
		const int nin = e->istream.channels, nout = e->ostream.channels;
		DepMap dm;
		if (query_deps(e, dm, max_in, max_out)) {
			const std::vector<ssize_t> t(samples);
			for (int o = 0; o < nout; ++o) {
				ssize_t d = 0;
				for (int k = 0; k < nin; ++k) if (dm.deps[o][k]) d = std::max(d, t[k]);
				samples[o] = d;
			}
		}
		else if (!(e->flags & (EFFECT_FLAG_CH_DEPS_IDENTITY | EFFECT_FLAG_OPT_REORDERABLE)) && nin != nout) {
			ssize_t d = 0;
			for (int k = 0; k < nin; ++k) d = std::max(d, samples[k]);
			for (int o = 0; o < nout; ++o) samples[o] = d;
		}
		if (e->drain_samples) e->drain_samples(e, samples.data());
		else if (e->ostream.fs != e->istream.fs) {
			const int g = gcd_i(e->ostream.fs, e->istream.fs);
			for (int o = 0; o < nout; ++o) samples[o] = mult_ceil(samples[o], e->ostream.fs / g, e->istream.fs / g);
		}
		for (int k = nout; k < nin; ++k) samples[k] = 0;
	}
	struct effect *head = plan.effects.front(), *tail = plan.effects.back();
	ssize_t d = 0;
	for (int o = 0; o < tail->ostream.channels; ++o) d = std::max(d, samples[o]);
	if (head->istream.fs != tail->ostream.fs) {
		const int g = gcd_i(head->istream.fs, tail->ostream.fs);
		d = (ssize_t) ((long long) d * (head->istream.fs / g) / (tail->ostream.fs / g));
	}
	plan.drain_frames = d;
	log_msg(LL_VERBOSE, "info: input drain frames: %zd", d);
}

bool build_chain(const char *chain_str, int fs, int channels, const char *dir, ChainPlan &plan)
{
	if (fs <= 0 || channels <= 0) { set_error("chain: error: invalid stream format"); return false; }
	plan.istream = { fs, channels };
	stream_info stream = plan.istream;
	const std::vector<Token> toks = lex(chain_str ? chain_str : "");
	Parser P{ &plan, &stream, dir, &toks };
	const size_t end = parse_tokens(P, 0, Selector(channels, 1), false, 1);
	if (end == (size_t) -1) return false;
	if (end != toks.size()) { set_error("chain: error: unexpected token: %s", toks[end].str.c_str()); return false; }
	plan.ostream = stream;
	optimize(plan);
	relink(plan);
	for (struct effect *e : plan.effects)
		if (e->prepare && e->prepare(e)) return false;
	if (!align_channels(plan)) return false;
	relink(plan);
	compute_drain(plan);
	return true;
}

}  // namespace dspamd
