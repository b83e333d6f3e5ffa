// Lean FP64 device math for the gfx950 MCMC kernels.
//
// Measured on MI355X (tools/ubench_isa.hip, cycles per wave-instruction at full occupancy):
// v_fma_f64 4.8, v_rcp/rsq_f64 16.3, IEEE a/b 61, ocml log 360, ocml sqrt 93, ocml sincospi 193,
// ocml exp 97.  The sampling kernels are bound by exactly these sequences, so the hot ones are
// restated here with <= 1-2 ulp error and without the range handling the kernels do not need:
//   fm_log      ~120 cycles   (fdlibm-style: frexp, s = f/(2+f), degree-7 polynomial in s^2)
//   fm_sqrt_rsq  ~55 cycles   (v_rsq_f64 + two coupled Newton steps; returns sqrt AND 1/sqrt)
//   fm_div       ~40 cycles   (v_rcp_f64 + two Newton steps, no final rounding fix-up)
//   fm_sincospi_quarter ~70   (|r| <= 1/4 only; the caller picks the quadrant from random bits)
// All results are compared with the CPU oracle to 1e-12 relative in tests/.
#pragma once
#include <hip/hip_runtime.h>

namespace mcmc {

// Single IEEE operations that the compiler may NOT contract into FMAs: the acceptance-rate
// recurrences of adaptMCMC feed discrete decisions (rate * i >= adapt_shape_start), so they
// must round exactly like the reference's separate multiply / add / divide.
__device__ __forceinline__ double mul_rn(double a, double b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double add_rn(double a, double b)
{
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ double div_rn(double a, double b)
{
#pragma clang fp contract(off)
    return a / b;
}

// A double constant delivered through a scalar register pair, re-materialized at every use (two s_mov_b32).
// Without it the compiler hoists the 64-bit polynomial coefficients of log / exp out of the sampling loops
// into VECTOR registers (an FP64 literal cannot be an instruction operand on gfx950) and then spills them:
// measured in rwm_pair.hip, 40 spilled VGPRs were coefficients.
__device__ __forceinline__ double sconst(double x)
{
    asm volatile("" : "+s"(x));
    return x;
}
// the same, but not before `after` has been computed: keeps the scalar pairs of a polynomial from all being
// live at once (the scheduler otherwise gathers them at the top of the block and scalar registers spill)
__device__ __forceinline__ double sconst(double x, double after)
{
    asm volatile("" : "+s"(x) : "v"(after));
    return x;
}

// ---- coefficients through wide scalar loads ---------------------------------------------------------------
// Two s_mov_b32 per FP64 coefficient make the SCALAR unit the busy one in polynomial-heavy code (PMC, C3:
// SQ_INSTS_SALU above SQ_INSTS_VALU).  A table in constant memory and one s_load_dwordx16 per eight coefficients
// take them off the instruction stream.
#define MCMC_CONST_AS __attribute__((address_space(4)))

// a * b + c with c read from a scalar register pair.  Written as the instruction itself: left to the compiler,
// fma(acc, x, <scalar>) becomes two v_mov_b32 plus a v_fmac_f64 whose addend must be a vector register.
__device__ __forceinline__ double fma_sc(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}

// a scalar register pair copied to a vector register pair in ONE instruction (the compiler emits two v_mov_b32)
__device__ __forceinline__ double vreg(double c)
{
    double d;
    asm("v_mov_b64 %0, %1" : "=v"(d) : "s"(c));
    return d;
}

// N (a multiple of 8) coefficients held in scalar registers
template <int N>
struct ScalarCoef {
    typedef double d8 __attribute__((ext_vector_type(8)));
    d8 v[N / 8];
    __device__ __forceinline__ double operator[](int k) const { return v[k >> 3][k & 7]; }
};
// ... fetched from a 64-byte aligned range of a __constant__ table; `t` must be wave-uniform
template <int N>
__device__ __forceinline__ ScalarCoef<N> load_coef(const MCMC_CONST_AS double *t)
{
    typedef typename ScalarCoef<N>::d8 d8;
    const MCMC_CONST_AS d8 *t8 = (const MCMC_CONST_AS d8 *)t;
    ScalarCoef<N> c;
#pragma unroll
    for (int b = 0; b < N / 8; ++b) c.v[b] = t8[b];
    return c;
}

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains vmcnt,
// i.e. it would hold the sampler until its trace stores to HBM are acknowledged, every iteration.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// a / b for finite, normal, well-scaled operands; error <= ~1.5 ulp.
__device__ __forceinline__ double fm_rcp(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    double e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    return r;
}
__device__ __forceinline__ double fm_div(double a, double b) { return a * fm_rcp(b); }

// sqrt(x) and 1/sqrt(x) for normal x > 0 (x = 0 gives NaN: callers test x > 0 first).
__device__ __forceinline__ void fm_sqrt_rsq(double x, double &s, double &rs)
{
    const double y = __builtin_amdgcn_rsq(x); // ~24 good bits
    double g = x * y;
    double h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); // ~48 bits
    h = fma(h, r, h);
    const double d = fma(-g, g, x);
    g = fma(d, h, g); // full precision
    r = fma(-h, g, 0.5);
    h = fma(h, r, h);
    s = g;
    rs = h + h;
}
__device__ __forceinline__ double fm_sqrt(double x)
{
    double s, rs;
    fm_sqrt_rsq(x, s, rs);
    return s;
}

// natural log.  FULL = true handles 0, +Inf, NaN and negatives like libm;
// FULL = false assumes a positive finite argument (uniforms in (0,1), tail probabilities).
template <bool FULL>
__device__ __forceinline__ double fm_log(double x)
{
    int e;
    double m = frexp(x, &e); // m in [0.5, 1); v_frexp_* normalise subnormals themselves
    const bool lo = m < sconst(0.70710678118654752440, m);
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double s = f * fm_rcp(2.0 + f);
    const double z = s * s;
    const double w = z * z;
    double t1 = fma(w, sconst(1.531383769920937332e-01, w), sconst(2.222219843214978396e-01, w));
    t1 = fma(w, t1, sconst(3.999999999940941908e-01, t1));
    t1 *= w;
    double t2 = fma(w, sconst(1.479819860511658591e-01, w), sconst(1.818357216161805012e-01, w));
    t2 = fma(w, t2, sconst(2.857142874366239149e-01, t2));
    t2 = fma(w, t2, sconst(6.666666666666735130e-01, t2));
    t2 *= z;
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    double res = dk * sconst(6.93147180369123816490e-01, dk) - ((hfsq - (s * (hfsq + R) + dk * sconst(1.90821492927058770002e-10, R))) - f);
    if (FULL) {
        if (x == 0.0) res = -__builtin_inf();
        if (x < 0.0 || x != x) res = __builtin_nan("");
        if (x == __builtin_inf()) res = x;
    }
    return res;
}

// exp(x): x = k ln2 + r, |r| <= ln2/2, degree-13 Taylor polynomial in r (truncation 4e-18), ldexp.
// <= 1 ulp on [-745, 709]; 0 / +Inf beyond, NaN propagates.  Coefficients come through scalar registers.
__device__ __forceinline__ double fm_exp(double x)
{
    const double xc = fmin(fmax(x, -1100.0), 1100.0); // keeps k inside int range; the result is 0 / Inf there anyway
    const double kd = rint(xc * sconst(1.4426950408889634074, xc));
    double r = fma(kd, -sconst(6.93147180369123816490e-01, kd), xc);
    r = fma(kd, -sconst(1.90821492927058770002e-10, r), r);
    double p = sconst(1.6059043836821613e-10, r);          // 1/13!
    p = fma(p, r, sconst(2.08767569878681e-09, p));          // 1/12!
    p = fma(p, r, sconst(2.505210838544172e-08, p));         // 1/11!
    p = fma(p, r, sconst(2.755731922398589e-07, p));         // 1/10!
    p = fma(p, r, sconst(2.7557319223985893e-06, p));        // 1/9!
    p = fma(p, r, sconst(2.48015873015873e-05, p));          // 1/8!
    p = fma(p, r, sconst(0.0001984126984126984, p));         // 1/7!
    p = fma(p, r, sconst(0.001388888888888889, p));          // 1/6!
    p = fma(p, r, sconst(0.008333333333333333, p));          // 1/5!
    p = fma(p, r, sconst(0.041666666666666664, p));          // 1/4!
    p = fma(p, r, sconst(0.16666666666666666, p));           // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    double res = ldexp(p, (int)kd);
    if (x != x) res = x;
    return res;
}

// sin(pi r), cos(pi r) for |r| <= 1/4 (near-minimax fits in r^2, max error 2.2e-16).
__device__ __forceinline__ void fm_sincospi_quarter(double r, double &s, double &c)
{
    const double t = r * r;
    double ps = sconst(0.000461531855383581, t);
    ps = fma(ps, t, sconst(-0.007370021586907771, ps));
    ps = fma(ps, t, sconst(0.08214586918000175, ps));
    ps = fma(ps, t, sconst(-0.5992645289396449, ps));
    ps = fma(ps, t, sconst(2.5501640398733763, ps));
    ps = fma(ps, t, sconst(-5.167712780049954, ps));
    ps = fma(ps, t, sconst(3.141592653589793, ps));
    double pc = sconst(-0.00010356747255199479, t);
    pc = fma(pc, t, sconst(0.0019294657440800042, pc));
    pc = fma(pc, t, sconst(-0.025806885652951306, pc));
    pc = fma(pc, t, sconst(0.23533063019088787, pc));
    pc = fma(pc, t, sconst(-1.3352627688519174, pc));
    pc = fma(pc, t, sconst(4.058712126416747, pc));
    pc = fma(pc, t, sconst(-4.934802200544679, pc));
    pc = fma(pc, t, 1.0);
    s = r * ps;
    c = pc;
}

// ---- fm_log<false>, fm_exp, fm_sincospi_quarter with their coefficients in scalar registers -----------------
// Same operations in the same order as the functions above (bit-identical results); the caller loads the ranges
// once (fm_tab) and may use them for many evaluations -- the generator wave of rwm_pair.hip: five Box-Muller
// pairs and a log per iteration from one load.
enum FmTab : int { FT_LOG = 0, FT_SINCOS = 16, FT_EXP = 32, FT_SIZE = 48 };
static __constant__ __attribute__((aligned(64))) double kFmTab[FT_SIZE] = {
    // FT_LOG: sqrt(1/2), the two interleaved polynomials, ln2 high / low
    0.70710678118654752440, 1.531383769920937332e-01, 2.222219843214978396e-01, 3.999999999940941908e-01,
    1.479819860511658591e-01, 1.818357216161805012e-01, 2.857142874366239149e-01, 6.666666666666735130e-01,
    6.93147180369123816490e-01, 1.90821492927058770002e-10, 0, 0, 0, 0, 0, 0,
    // FT_SINCOS: sin(pi r) / r (7, leading first), cos(pi r) (7, leading first; the constant term is 1)
    0.000461531855383581, -0.007370021586907771, 0.08214586918000175, -0.5992645289396449,
    2.5501640398733763, -5.167712780049954, 3.141592653589793, -0.00010356747255199479,
    0.0019294657440800042, -0.025806885652951306, 0.23533063019088787, -1.3352627688519174,
    4.058712126416747, -4.934802200544679, 0, 0,
    // FT_EXP: log2(e), ln2 high / low, 1/13! .. 1/3!
    1.4426950408889634074, 6.93147180369123816490e-01, 1.90821492927058770002e-10, 1.6059043836821613e-10,
    2.08767569878681e-09, 2.505210838544172e-08, 2.755731922398589e-07, 2.7557319223985893e-06,
    2.48015873015873e-05, 0.0001984126984126984, 0.001388888888888889, 0.008333333333333333,
    0.041666666666666664, 0.16666666666666666, 0, 0};

// N coefficients from `first`; `after` (wave-uniform) pins the loads behind its definition, i.e. inside the loop
template <int N>
__device__ __forceinline__ ScalarCoef<N> fm_tab(int first, int after)
{
    const MCMC_CONST_AS double *t = (const MCMC_CONST_AS double *)kFmTab + first;
    asm volatile("" : "+s"(t) : "s"(after));
    return load_coef<N>(t);
}

__device__ __forceinline__ double fm_log_t(double x, const ScalarCoef<16> &t)
{
    int e;
    double m = frexp(x, &e);
    const bool lo = m < t[0];
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double s = f * fm_rcp(2.0 + f);
    const double z = s * s;
    const double w = z * z;
    double t1 = fma_sc(w, vreg(t[1]), t[2]);
    t1 = fma_sc(w, t1, t[3]);
    t1 *= w;
    double t2 = fma_sc(w, vreg(t[4]), t[5]);
    t2 = fma_sc(w, t2, t[6]);
    t2 = fma_sc(w, t2, t[7]);
    t2 *= z;
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return dk * t[8] - ((hfsq - (s * (hfsq + R) + dk * t[9])) - f);
}

__device__ __forceinline__ double fm_exp_t(double x, const ScalarCoef<16> &t)
{
    const double xc = fmin(fmax(x, -1100.0), 1100.0);
    const double kd = rint(xc * t[0]);
    double r = fma(kd, -t[1], xc);
    r = fma(kd, -t[2], r);
    double p = vreg(t[3]);
#pragma unroll
    for (int k = 4; k < 14; ++k) p = fma_sc(p, r, t[k]);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    double res = ldexp(p, (int)kd);
    if (x != x) res = x;
    return res;
}

__device__ __forceinline__ void fm_sincospi_quarter_t(double r, double &s, double &c, const ScalarCoef<16> &t)
{
    const double x = r * r;
    double ps = vreg(t[0]);
#pragma unroll
    for (int k = 1; k < 7; ++k) ps = fma_sc(ps, x, t[k]);
    double pc = vreg(t[7]);
#pragma unroll
    for (int k = 8; k < 14; ++k) pc = fma_sc(pc, x, t[k]);
    pc = fma(pc, x, 1.0);
    s = r * ps;
    c = pc;
}

} // namespace mcmc
