// thiran_roots.cpp -- the poles of a Thiran all-pass of order n (the reference's fractional delay, allpass.h:73-118: any order up to 50 through
// delay.c:700-701's range check) in 113-bit arithmetic.  Compiled by the host compiler ($(HOSTCXX), not as HIP: __float128 does not exist in the
// device pass); no libquadmath calls: sums, products and quotients only.  Where the host compiler has no __float128 the same code runs in long
// double and the final check (the double-rounded sections multiplied back against the polynomial, 1e-13) refuses the orders it cannot serve.
// The interface is plain C (an array of doubles): this object is the one translation unit built by another compiler than the rest.
//
// The reference runs the filter as the ladder of Koshita et al.; this backend runs it as second-order sections in the fused cascade kernels, which
// needs the roots of  sum_k a_k z^(n-k),  a_k = (-1)^k C(n, k) prod_{i=0..n} (D - n + i) / (D - n + k + i).  In long double the roots of orders
// from about 34 on no longer reproduce the polynomial to 1e-13 (they cluster); with 113 bits they do to 1e-16 up to order 50 and the double-rounded
// sections' response is within 1e-15 of z^-n A(1/z) / A(z) over the whole band.
#include <vector>
#include <algorithm>

namespace {
#ifdef __SIZEOF_FLOAT128__
typedef __float128 qf;
#else
typedef long double qf;
#endif
struct cq { qf re, im; };
inline cq operator+(cq a, cq b) { return { a.re + b.re, a.im + b.im }; }
inline cq operator-(cq a, cq b) { return { a.re - b.re, a.im - b.im }; }
inline cq operator*(cq a, cq b) { return { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; }
inline cq operator/(cq a, cq b) { const qf d = b.re * b.re + b.im * b.im; return { (a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d }; }
inline qf absq(qf x) { return x < 0 ? -x : x; }
}

// out: [cap][5] doubles (b0, b1, b2, a1, a2 of each section); returns the number of sections written, or -1
extern "C" __attribute__((visibility("hidden"))) int dspamd_thiran_pole_sections(int n, double D, double *out, int cap)
{
	if (n < 1 || n > 64 || !out || cap < (n + 1) / 2) return -1;
	int n_out = 0;
	std::vector<qf> a(n + 1);
	for (int k = 0; k <= n; ++k) {
		qf c = 1;
		for (int i = 1; i <= k; ++i) c = c * (qf) (n - k + i) / (qf) i;      // C(n, k)
		qf pr = 1;
		for (int i = 0; i <= n; ++i) pr *= ((qf) D - n + i) / ((qf) D - n + k + i);
		a[k] = ((k & 1) ? -c : c) * pr;
	}
	auto P = [&](cq z) { cq v = { a[0], 0 }; for (int k = 1; k <= n; ++k) v = v * z + cq{ a[k], 0 }; return v; };
	auto dP = [&](cq z) { cq v = { (qf) n * a[0], 0 }; for (int k = 1; k < n; ++k) v = v * z + cq{ (qf) (n - k) * a[k], 0 }; return v; };
	// Durand-Kerner from a spiral inside the unit circle, then Newton on each root
	std::vector<cq> r(n);
	{
		const cq g = { (qf) 0.4, (qf) 0.9 };
		cq cur = { (qf) 0.6, 0 };
		for (int i = 0; i < n; ++i) { r[i] = cur; cur = cur * g; }
	}
	for (int it = 0; it < 400; ++it) {
		qf move = 0;
		for (int i = 0; i < n; ++i) {
			cq den = { a[0], 0 };
			for (int j = 0; j < n; ++j) if (j != i) den = den * (r[i] - r[j]);
			const cq d = P(r[i]) / den;
			r[i] = r[i] - d;
			move = std::max(move, std::max(absq(d.re), absq(d.im)));
		}
		if (move < (qf) 1e-29) break;
	}
	for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) r[i] = r[i] - P(r[i]) / dP(r[i]);
	// complex roots with their conjugates, real ones two by two
	std::vector<cq> up;
	std::vector<qf> real;
	for (const cq &z : r) {
		if (!(z.re * z.re + z.im * z.im < 1)) return -1;
		if (absq(z.im) < (qf) 1e-20) real.push_back(z.re);
		else if (z.im > 0) up.push_back(z);
	}
	if ((int) (2 * up.size() + real.size()) != n) return -1;
	std::vector<qf> rec(1, (qf) 1);                                    // product of the sections AS ROUNDED to double, for the check
	auto section = [&](qf c1q, qf c2q, int deg) {
		const double c1 = (double) c1q, c2 = (double) c2q;
		double *o = out + 5 * n_out++;
		if (deg == 2) { o[0] = c2; o[1] = c1; o[2] = 1.0; o[3] = c1; o[4] = c2; }      // (c2 + c1 z^-1 + z^-2) / (1 + c1 z^-1 + c2 z^-2)
		else { o[0] = c1; o[1] = 1.0; o[2] = 0.0; o[3] = c1; o[4] = 0.0; }             // (c1 + z^-1) / (1 + c1 z^-1)
		std::vector<qf> q(rec.size() + deg, (qf) 0);
		for (size_t i = 0; i < rec.size(); ++i) { q[i] += rec[i]; q[i + 1] += rec[i] * (qf) c1; if (deg == 2) q[i + 2] += rec[i] * (qf) c2; }
		rec.swap(q);
	};
	for (const cq &z : up) section(-2 * z.re, z.re * z.re + z.im * z.im, 2);
	std::sort(real.begin(), real.end());
	for (size_t i = 0; i + 1 < real.size(); i += 2) section(-(real[i] + real[i + 1]), real[i] * real[i + 1], 2);
	if (real.size() & 1) section(-real.back(), 0, 1);
	qf err = 0, big = 0;
	for (int k = 0; k <= n; ++k) { err = std::max(err, absq(rec[k] - a[k])); big = std::max(big, absq(a[k])); }
	return err <= (qf) 1e-13 * big ? n_out : -1;
}
