// Adaptive random-walk Metropolis for 12 < d <= 64: one chain per ROW of 16 (d <= 16), 32 (d <= 32) or 64 lanes (gfx950).
//
// Replaces the main loop of adaptMCMC / rwMCMC (adaptMCMC_source.cpp:108-216, :252-313) in the range where the chain
// state no longer fits one lane (the factor R and covmat_empirical are 136 doubles each at d = 16, 528 at d = 32) but
// a 256-thread workgroup per chain leaves > 90 % of the lanes idle (d = 16 ran 60 x slower than d = 10).
//
// Layout.  Lane j of a row owns COORDINATE j of its chain: theta_j, theta_mean_j, Y_j, and COLUMN j of the two
// matrices -- R_kj (upper factor, zero below the diagonal) and covmat_empirical_kj (the full symmetric column) -- in
// L registers each, statically indexed.  Four chains (d <= 16) or two (d <= 32) share a wavefront.  Everything a
// chain needs from "the other coordinates" is one cross-lane primitive:
//     v_fmac_f64_dpp acc, x, y row_newbcast:k      acc += (x of lane k of this 16-lane row) * y
// one instruction at the FP64 FMA rate (tools/ubench_r2.hip), so a matrix-vector product is L instructions:
//     proposal   theta_j + s sum_k Y_k R_kj                      (:153-154, terms in the reference's order)
//     target     -0.5 sum_j v_j sum_k H_jk v_k                   (the row of H from LDS, shared by all chains)
//     update     C_kj = C_kj (i-1)/i + ((i-1)/i^2 r_k) r_j       (update_sigma, :42-45)
//     Cholesky   right-looking, in place in the R registers: at pivot p every lane scales its entry of row p, and
//                the trailing update a_ij -= R_pi R_pj is one such instruction per remaining row i
// For 32-lane rows the operand is first spread over both 16-lane halves by v_permlane16_swap_b32 (two instructions
// give "lanes 0-15 of the chain in both halves" and "lanes 16-31 in both halves"); for 64-lane rows -- one chain per
// wavefront -- v_permlane32_swap_b32 adds the other half, and covmat_empirical moves to LDS (see RowLayout).  Scalars of a chain (acceptance
// rate, scaling_sd, target value, ...) are computed redundantly, bit-identically, in every lane of its row.
//
// chol(s^2 Sigma) = s chol(Sigma): as in the other kernels R is the UNSCALED factor and the scale is kept apart,
// so size-phase iterations do no factorization.
//
// Trace: lane j holds coordinate j, so a store instruction of a wave would touch 8 bytes in each of 16 x 4 lines.
// The 16 (8) chains of a workgroup are consecutive; each kept iteration goes through a double-buffered LDS tile
// [k][chain] and leaves as whole 128-byte (64-byte) runs, one LDS-only barrier per kept iteration.
//
// Serves Philox runs without per-iteration diagnostics (the per-workgroup kernel of rwm_block.hip carries injected
// streams, the sigma_empirical trace and diagnostics for d > 12); Gaussian target.
#include "accept_window.h"
#include "common.h"
#include "philox.h"
#include "targets.h"

namespace mcmc {

template <int V>
struct RowIC {
    static constexpr int value = V;
};
template <int K, int N, class F>
__device__ __forceinline__ void row_unroll(F &&f)
{
    if constexpr (K < N) {
        f(RowIC<K>{});
        row_unroll<K + 1, N>(f);
    }
}

// A VGPR written by a VALU instruction must not be read through DPP for two wait states.  The compiler keeps that
// rule for its own DPP instructions but cannot see into the asm below, so every value that is about to become a DPP
// operand passes through this fence once: the s_nop sits behind the instruction that produced x and ahead of every
// later use of it.  Nothing in the language ties the fence to the asm blocks below -- a register copy the compiler
// inserts between the two (a live-range split, an AGPR read under pressure) would be read too early -- so the BUILD pins
// it: tools/check_dpp_hazard.py disassembles the code object and tests/test_abi_and_host.py fails if any DPP instruction
// reads a VGPR that a VALU instruction wrote within the two preceding wait states.
__device__ __forceinline__ double dpp_ready(double x)
{
    asm volatile("s_nop 1" : "+v"(x));
    return x;
}

// acc += (x of lane K of this lane's 16-lane row) * y      (x: a dpp_ready value)
template <int K>
__device__ __forceinline__ void fmac_rb(double &acc, double x, double y)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(K));
}
// The same instruction in blocks, one asm statement each: the compiler closes every asm statement with an s_nop (it
// cannot see what the last instruction inside was), which for single-instruction statements doubled the count.
// acc += sum_k (x of lane 8 G + k of the row) * m[k], k = 0..7, in that order
template <int G>
__device__ __forceinline__ void fmac_dot8(double &acc, double x, const double *m);
template <>
__device__ __forceinline__ void fmac_dot8<0>(double &acc, double x, const double *m)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(acc)
        : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]));
}
template <>
__device__ __forceinline__ void fmac_dot8<1>(double &acc, double x, const double *m)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %3 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %4 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %5 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %6 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %7 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %8 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %9 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(acc)
        : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]));
}
// a[k] += (x of lane k of the row) * y, k = 0..15
__device__ __forceinline__ void fmac_rows16(double *a, double x, double y)
{
    asm("v_fmac_f64_dpp %0, %16, %17 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %16, %17 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %16, %17 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %16, %17 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %4, %16, %17 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %5, %16, %17 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %6, %16, %17 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %7, %16, %17 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %8, %16, %17 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %9, %16, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %10, %16, %17 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %11, %16, %17 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %12, %16, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %13, %16, %17 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %14, %16, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %15, %16, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
        : "v"(x), "v"(y));
}
// a[4 G + q] += (x of lane 4 G + q of the row) * y, q = 0..3
template <int G>
__device__ __forceinline__ void fmac_rows4(double *a, double x, double y);
template <>
__device__ __forceinline__ void fmac_rows4<0>(double *a, double x, double y)
{
    asm("v_fmac_f64_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %4, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])
        : "v"(x), "v"(y));
}
template <>
__device__ __forceinline__ void fmac_rows4<1>(double *a, double x, double y)
{
    asm("v_fmac_f64_dpp %0, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %4, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
        : "v"(x), "v"(y));
}
template <>
__device__ __forceinline__ void fmac_rows4<2>(double *a, double x, double y)
{
    asm("v_fmac_f64_dpp %0, %4, %5 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %5 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %4, %5 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %4, %5 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11])
        : "v"(x), "v"(y));
}
template <>
__device__ __forceinline__ void fmac_rows4<3>(double *a, double x, double y)
{
    asm("v_fmac_f64_dpp %0, %4, %5 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %5 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %4, %5 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %4, %5 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
        : "v"(x), "v"(y));
}
// x of lane K of this lane's 16-lane row
// (bound_ctrl is set on every DPP move of this file: under their controls -- row_newbcast, quad_perm, row mirrors -- every
// lane has a source, and with it the compiler does not materialise an "old" value per move: two v_mov_b32 each)
template <int K>
__device__ __forceinline__ double row_bcast(double x)
{
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x150 + K, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x150 + K, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// A value per lane of a chain's row, made available to every lane of the row: for 16-lane rows the value itself, for
// 32-lane rows two registers, `lo` = the values of lanes 0-15 of the chain in both halves, `hi` = those of lanes 16-31.
template <int L>
struct RowVec;
template <>
struct RowVec<16> {
    double lo;
    __device__ __forceinline__ explicit RowVec(double x) : lo(dpp_ready(x)) {}
    template <int K>
    __device__ __forceinline__ void fmac(double &acc, double y) const { fmac_rb<K>(acc, lo, y); }
    template <int K>
    __device__ __forceinline__ double get() const { return row_bcast<K>(lo); }
    // acc += sum_k v_k m[k] over the whole row, k ascending
    template <int G>
    __device__ __forceinline__ void dot8(double &acc, const double *m) const { fmac_dot8<G>(acc, lo, m); } // terms 8 G .. 8 G + 7
    // a[k] += v_k y for every k
    template <int K>
    __device__ __forceinline__ void rows(double (&a)[K], double y) const { fmac_rows16(a, lo, y); }
    // a[k] += v_k y for k >= 4 (P / 4): the rows of the group that holds P and all later ones
    template <int P, int K>
    __device__ __forceinline__ void rows_from(double (&a)[K], double y) const
    {
        if constexpr (P / 4 <= 0) fmac_rows4<0>(a, lo, y);
        if constexpr (P / 4 <= 1) fmac_rows4<1>(a, lo, y);
        if constexpr (P / 4 <= 2) fmac_rows4<2>(a, lo, y);
        if constexpr (P / 4 <= 3) fmac_rows4<3>(a, lo, y);
    }
};
template <>
struct RowVec<32> {
    double lo, hi;
    __device__ __forceinline__ explicit RowVec(double x)
    {
        const uint64_t u = (uint64_t)__double_as_longlong(x);
        const auto a = __builtin_amdgcn_permlane16_swap((uint32_t)u, (uint32_t)u, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap((uint32_t)(u >> 32), (uint32_t)(u >> 32), false, false);
        lo = __longlong_as_double((long long)(((uint64_t)b[0] << 32) | a[0]));
        hi = __longlong_as_double((long long)(((uint64_t)b[1] << 32) | a[1]));
        asm volatile("s_nop 1" : "+v"(lo), "+v"(hi));
    }
    template <int K>
    __device__ __forceinline__ void fmac(double &acc, double y) const
    {
        if constexpr (K < 16)
            fmac_rb<K>(acc, lo, y);
        else
            fmac_rb<K - 16>(acc, hi, y);
    }
    template <int K>
    __device__ __forceinline__ double get() const
    {
        if constexpr (K < 16)
            return row_bcast<K>(lo);
        else
            return row_bcast<K - 16>(hi);
    }
    template <int G>
    __device__ __forceinline__ void dot8(double &acc, const double *m) const
    {
        if constexpr (G < 2)
            fmac_dot8<G>(acc, lo, m);
        else
            fmac_dot8<G - 2>(acc, hi, m);
    }
    // K = number of matrix rows the kernel keeps (32, or 24 when d <= 24)
    template <int K>
    __device__ __forceinline__ void rows(double (&a)[K], double y) const
    {
        fmac_rows16(a, lo, y);
        if constexpr (K == 32) {
            fmac_rows16(a + 16, hi, y);
        } else {
            fmac_rows4<0>(a + 16, hi, y);
            fmac_rows4<1>(a + 16, hi, y);
        }
    }
    template <int P, int K>
    __device__ __forceinline__ void rows_from(double (&a)[K], double y) const
    {
        if constexpr (P / 4 <= 0) fmac_rows4<0>(a, lo, y);
        if constexpr (P / 4 <= 1) fmac_rows4<1>(a, lo, y);
        if constexpr (P / 4 <= 2) fmac_rows4<2>(a, lo, y);
        if constexpr (P / 4 <= 3) fmac_rows4<3>(a, lo, y);
        if constexpr (P / 4 <= 4 && K > 16) fmac_rows4<0>(a + 16, hi, y);
        if constexpr (P / 4 <= 5 && K > 20) fmac_rows4<1>(a + 16, hi, y);
        if constexpr (P / 4 <= 6 && K > 24) fmac_rows4<2>(a + 16, hi, y);
        if constexpr (P / 4 <= 7 && K > 28) fmac_rows4<3>(a + 16, hi, y);
    }
};

// 64-lane rows (one chain per wavefront): four registers, v[r] = the values of lanes 16 r .. 16 r + 15 of the chain in
// every 16-lane row -- v_permlane16_swap_b32 pairs the rows inside each half, v_permlane32_swap_b32 the halves.
template <>
struct RowVec<64> {
    double v[4];
    __device__ __forceinline__ explicit RowVec(double x)
    {
        const uint64_t u = (uint64_t)__double_as_longlong(x);
        uint32_t w[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t t = (uint32_t)(u >> (32 * h));
            const auto a = __builtin_amdgcn_permlane16_swap(t, t, false, false); // a[0] = rows (0,0,2,2), a[1] = rows (1,1,3,3)
            const auto e = __builtin_amdgcn_permlane32_swap(a[0], a[0], false, false); // e[0] = rows (0,0,0,0), e[1] = (2,2,2,2)
            const auto o = __builtin_amdgcn_permlane32_swap(a[1], a[1], false, false); // o[0] = rows (1,1,1,1), o[1] = (3,3,3,3)
            w[h][0] = e[0];
            w[h][1] = o[0];
            w[h][2] = e[1];
            w[h][3] = o[1];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = __longlong_as_double((long long)(((uint64_t)w[1][r] << 32) | w[0][r]));
        asm volatile("s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
    }
    template <int K>
    __device__ __forceinline__ void fmac(double &acc, double y) const { fmac_rb<K % 16>(acc, v[K / 16], y); }
    template <int K>
    __device__ __forceinline__ double get() const { return row_bcast<K % 16>(v[K / 16]); }
    template <int G>
    __device__ __forceinline__ void dot8(double &acc, const double *m) const { fmac_dot8<G % 2>(acc, v[G / 2], m); }
    template <int K>
    __device__ __forceinline__ void rows(double (&a)[K], double y) const
    {
        fmac_rows16(a, v[0], y);
        fmac_rows16(a + 16, v[1], y);
        fmac_rows16(a + 32, v[2], y);
        if constexpr (K == 64) fmac_rows16(a + 48, v[3], y);
    }
    // a[16 B + k] += v_{16 B + k} y, k = 0..15: one block of sixteen rows
    template <int B>
    __device__ __forceinline__ void rows16(double *a16, double y) const { fmac_rows16(a16, v[B], y); }
    // rows P .. d - 1 (whole groups of four; whole blocks of sixteen are skipped once they lie past d)
    template <int P, int K>
    __device__ __forceinline__ void rows_from(double (&a)[K], double y, int d) const
    {
        row_unroll<0, K / 16>([&](auto bc) {
            constexpr int B = decltype(bc)::value;
            if constexpr (P / 16 <= B) {
                if (16 * B < d) { // uniform
                    row_unroll<0, 4>([&](auto gc) {
                        constexpr int G = 4 * B + decltype(gc)::value;
                        if constexpr (P / 4 <= G) fmac_rows4<G % 4>(a + 16 * B, v[B], y);
                    });
                }
            }
        });
    }
};

// sum over the lanes of a chain's row, the same bits in every lane: four DPP steps inside a 16-lane row (the pairs are
// symmetric, and IEEE addition is commutative), then the two halves of a 32-lane row / the four rows of a 64-lane one
template <int L>
__device__ __forceinline__ double row_sum(double x)
{
    auto dpp_add = [](double y, auto ctrl) -> double {
        constexpr int C = decltype(ctrl)::value;
        const uint64_t b = (uint64_t)__double_as_longlong(y);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, C, 0xf, 0xf, true);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), C, 0xf, 0xf, true);
        return y + __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
    };
    x = dpp_add(x, RowIC<0xB1>{});  // quad_perm [1,0,3,2]: lane ^ 1
    x = dpp_add(x, RowIC<0x4E>{});  // quad_perm [2,3,0,1]: lane ^ 2
    x = dpp_add(x, RowIC<0x141>{}); // row_half_mirror: the other quad of the same half row
    x = dpp_add(x, RowIC<0x140>{}); // row_mirror: the other half of the row
    if constexpr (L == 32) {
        const RowVec<32> v(x);
        x = v.lo + v.hi;
    }
    if constexpr (L == 64) {
        const RowVec<64> v(x);
        x = (v.v[0] + v.v[1]) + (v.v[2] + v.v[3]);
    }
    return x;
}

// parameter block (doubles), staged in LDS: min diag(covmat), prod diag(R0), -, -, tmean[L], init[L],
// Hs[k][j] = (H_jk + H_kj) / 2 (zero beyond d), R0c[k][j] = chol(covmat)_kj for k <= j < d (zero elsewhere)
//
// L = 64 (one chain per wavefront, 32 < d <= 64): the R columns alone are 128 registers, so covmat_empirical lives in
// LDS, per chain, as block columns: lane j = 16 m + q keeps rows 0 .. 16 (m + 1) - 1 of its column (the upper triangle
// rounded up to whole 16-row blocks; the few entries below the diagonal are the symmetric ones and are kept up to date
// like the rest), row k at block m's base + 16 k + q -- consecutive lanes, consecutive doubles.  256 m (m + 1) / 2 doubles
// up to block m: 20 KB per chain at d = 64.  chol(covmat) stays in HBM / L2 (read at start and when the size phase
// comes back), the workgroup is as many wavefronts as the CU's LDS holds chains (6 at d = 64, 8 up to d = 48).
template <int L>
struct RowLayout {
    static constexpr int MINVAR = 0, PROD0 = 1, TMEAN = 4, INIT = 4 + L, HS = 4 + 2 * L, R0 = HS + L * L, TOTAL = R0 + L * L;
    static constexpr int STAGED = L == 64 ? R0 : TOTAL; // the part of the block every workgroup copies into LDS
    static constexpr int CPW = 256 / L; // chains per workgroup (L = 64: chosen at launch)
    static constexpr int csize(int d) { return L == 64 ? 128 * ((d + 15) / 16) * ((d + 15) / 16 + 1) : 0; } // covmat_empirical of one chain
    static constexpr int lds_doubles(int d, int cpw) { return STAGED + cpw * csize(d) + 2 * d * cpw; }
};

// K: the matrix rows a lane keeps of its column -- L, or 3 L / 4 for d <= 24 (L = 32) and d <= 48 (L = 64): a quarter fewer
// registers and a quarter fewer FMAs in every dot product, update and pivot step (d = 17..31 and
// 33..47 idled a third to a half of the work of their row)
template <int L, int K, bool STATE>
__global__ __launch_bounds__(L == 64 ? 512 : 256, L == 16 ? 4 : (L == 32 ? 2 : 1)) void rwm_row_kernel(const RwmArgs a, const double *__restrict__ prm)
{
    static_assert(K == L || (L >= 32 && K == 3 * L / 4), "K is L or three quarters of it");
    using Lay = RowLayout<L>;
    constexpr int RPW = 64 / L;
    constexpr bool CL = L == 64; // covmat_empirical in LDS
    extern __shared__ double lds[];
    const int NTHR = CL ? (int)blockDim.x : 256, CPW = CL ? (int)(blockDim.x >> 6) : Lay::CPW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & (L - 1);       // this lane's coordinate
    const int cl = wave * RPW + lane / L; // this lane's chain inside the workgroup
    const int d = a.d;
    const bool act = j < d;
    const int64_t n = a.n_chains;
    const int64_t c0 = (int64_t)blockIdx.x * CPW, c = c0 + cl;
    const bool live = c < n;
    const int64_t cc = live ? c : n - 1; // rows past the end shadow the last chain and store nothing
    const uint64_t gchain = (uint64_t)(a.chain_offset + cc);
    const PhiloxKey key(a.seed);
    const int ass = a.adapt_size_start, ashs = a.adapt_shape_start;
    const int window = a.acceptance_window;
    const double wgt = a.acceptance_rate_weight;
    const bool need_moments = ashs != 0 || a.d_final_mean || a.d_final_cov || (STATE && (a.d_state_in || a.d_state_out));
    const int i0 = STATE ? a.iter_offset : 0;

    for (int e = tid; e < Lay::STAGED; e += NTHR) lds[e] = prm[e];
    const int MB = (d + 15) >> 4, mj = j >> 4; // 16-row blocks of the matrix, this lane's block
    // this chain's covmat_empirical (L = 64): entry (k, j) at Cl[16 k] for k < 16 (mj + 1)
    double *const Cl = lds + Lay::STAGED + (CL ? cl * Lay::csize(d) + 128 * mj * (mj + 1) + (j & 15) : 0);
    if constexpr (CL) {
        for (int e = lane; e < Lay::csize(d); e += 64) lds[Lay::STAGED + cl * Lay::csize(d) + e] = 0.0;
    }
    __syncthreads();
    const double *const Hs = lds + Lay::HS + j;   // H_jk at Hs[k * L]
    const double *const R0c = (CL ? prm : lds) + Lay::R0 + j; // chol(covmat)_kj at R0c[k * L]
    double *const stage = lds + Lay::STAGED + CPW * Lay::csize(d); // [2][d][CPW] trace tile
    const double tm = lds[Lay::TMEAN + j];
    const double min_var = lds[Lay::MINVAR], prodR0 = lds[Lay::PROD0];

    double Rc[K], Cc[CL ? 1 : K];
#pragma unroll
    for (int k = 0; k < K; ++k) Rc[k] = R0c[k * L];
#pragma unroll
    for (int k = 0; k < (CL ? 1 : K); ++k) Cc[k] = 0.0;
    // one 16-row block B of this lane's column of covmat_empirical at a time (L = 64): f(k, reference to the entry)
    auto c_block = [&](auto bc, auto &&f) {
        constexpr int B = decltype(bc)::value;
        if (B <= mj && mj < MB) { // the lanes whose column reaches into block B (rows 16 B .. 16 B + 15): whole 16-lane rows
            double t[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) t[k] = Cl[16 * (16 * B + k)];
            f(t);
#pragma unroll
            for (int k = 0; k < 16; ++k) Cl[16 * (16 * B + k)] = t[k];
        }
    };
    double theta = act ? (a.d_init ? a.d_init[(int64_t)j * n + cc] : lds[Lay::INIT + j]) : tm; // padding: v = theta - m = 0
    double mean = theta;

    // -0.5 (x - m)^T H (x - m), the same bits in every lane of the row
    auto target = [&](double x) -> double {
        const double v = x - tm;
        const RowVec<L> vv(v);
        double g = 0.0;
        // eight terms per block: the row of H comes from LDS eight values at a time, ONE BLOCK AHEAD of its use (the
        // scheduling fence keeps the requests of block G + 1 in front of the FMAs of block G; read where they are used,
        // every block was an exposed LDS round trip)
        if constexpr (L == 16) { // 128 registers per lane: no room for a second block (measured: 36 more spills)
            row_unroll<0, K / 8>([&](auto gc) {
                constexpr int G = decltype(gc)::value;
                double h1[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) h1[k] = Hs[(8 * G + k) * L];
                vv.template dot8<G>(g, h1);
            });
            return -0.5 * row_sum<L>(v * g);
        }
        double h[2][8];
#pragma unroll
        for (int k = 0; k < 8; ++k) h[0][k] = Hs[k * L];
        row_unroll<0, K / 8>([&](auto gc) {
            constexpr int G = decltype(gc)::value;
            if constexpr (G + 1 < K / 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) h[(G + 1) & 1][k] = Hs[(8 * (G + 1) + k) * L];
            }
            __builtin_amdgcn_sched_barrier(0);
            vv.template dot8<G>(g, h[G & 1]);
            __builtin_amdgcn_sched_barrier(0);
        });
        return -0.5 * row_sum<L>(v * g);
    };

    double prodR = prodR0;
    bool base_emp = false; // R holds chol(covmat_empirical)
    double s_eff = 1.0;    // covmat_proposal = s_eff^2 R^T R
    double rootTerm, sq_unused;
    fm_sqrt_rsq(a.twopi_pow_d * (prodR * prodR), sq_unused, rootTerm);
    double t_cur = target(theta);
    double rate = 0.0, scaling_sd = 1.0;
    AcceptWindow win;
    int win_pos = accept_window_pos(i0, window);
    unsigned long long *const ring = a.d_win_ring ? a.d_win_ring + cc : nullptr;
    int32_t status = MCMC_CHAIN_OK, n_shape = 0;
    bool dead = false;
    const RwmStateLayout SL = rwm_state_layout(d);
    if (STATE && a.d_state_in) { // resume: everything the loop carries from one iteration to the next
        const double *const sp = a.d_state_in + cc;
        if (act) {
            theta = sp[(int64_t)(SL.THETA + j) * n];
            mean = sp[(int64_t)(SL.MEAN + j) * n];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (act && k < d) { // column j of the symmetric covmat_empirical from its packed upper triangle; of R the part k <= j
                const double cv = sp[(int64_t)(SL.C + (k <= j ? tri(k, j) : tri(j, k))) * n];
                if constexpr (CL) {
                    if (k < 16 * (mj + 1)) Cl[16 * k] = cv;
                } else {
                    Cc[k] = cv;
                }
                Rc[k] = k <= j ? sp[(int64_t)(SL.R + tri(k, j)) * n] : 0.0;
            }
        }
        const double *const ss = sp + (int64_t)SL.S * n;
        t_cur = ss[ST_TCUR * n];
        rate = ss[ST_RATE * n];
        scaling_sd = ss[ST_SD * n];
        s_eff = ss[ST_SEFF * n];
        prodR = ss[ST_PRODR * n];
        rootTerm = ss[ST_ROOT * n];
        win.bits = ((uint64_t)(uint32_t)ss[ST_WHI * n] << 32) | (uint64_t)(uint32_t)ss[ST_WLO * n];
        win.len = (int)ss[ST_WLEN * n];
        win.count = (int)ss[ST_WCOUNT * n];
        if (live && j == 0)
            for (int wd = 0; wd < a.win_words; ++wd)
                ring[(int64_t)wd * n] = ((uint64_t)(uint32_t)sp[(int64_t)(SL.TOTAL + 2 * wd + 1) * n] << 32) |
                                        (uint64_t)(uint32_t)sp[(int64_t)(SL.TOTAL + 2 * wd) * n];
        base_emp = ss[ST_BASE * n] != 0.0;
        status = (int32_t)ss[ST_STATUS * n];
        n_shape = (int32_t)ss[ST_NSHAPE * n];
        dead = status != MCMC_CHAIN_OK;
        if (ss[ST_TAG * n] != (double)MCMC_KERNEL_ROW || ss[ST_ITER * n] != (double)i0) { // not a state this launch can continue
            dead = true;
            status = ((i0 + 1) << 4) | MCMC_CHAIN_BAD_STATE;
            theta = __builtin_nan("");
        }
    }

    int next_keep = (i0 / a.thin + 1) * a.thin; // next iteration whose state is stored
    int64_t t_kept = 0;
    for (int i = i0 + 1; i <= i0 + a.n_iterations; ++i) {
        const int li = i - i0; // position inside this launch (cooling table)
        const double di = (double)i;
        // ---- draws: this lane's normal, the chain's uniform (:149-152,174) ----------------------------------
        // ONE Philox block per lane: the even lane 2 b draws block b (normals 2 b, 2 b + 1) and hands the second normal
        // to its odd neighbour, which leaves the odd lanes free -- lane 1 of every 16-lane row draws the uniform's block
        // with the same instructions (log u0 is Box-Muller's first step) and its row reads log A from it.
        double Y, logA;
        {
            PhiloxKey kk = key; // opaque copy: the ten round keys are formed by scalar adds here, not hoisted and spilled
            asm volatile("" : "+s"(kk.k0), "+s"(kk.k1));
            const ScalarCoef<16> tl = fm_tab<16>(FT_LOG, i), ts = fm_tab<16>(FT_SINCOS, i);
            const bool uni = (j & 15) == 1;
            double z0, z1, lg;
            philox_normal2_log_t(kk, gchain, (uint32_t)i, uni ? TAG_RWM_UNIFORM : TAG_RWM_NORMAL, uni ? 0u : (uint32_t)(j >> 1), z0, z1, lg,
                                 tl, ts);
            const uint64_t zb = (uint64_t)__double_as_longlong(z1); // quad_perm [0,0,2,2]: the even neighbour's second normal
            const uint32_t zl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)zb, 0xA0, 0xf, 0xf, true);
            const uint32_t zh = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(zb >> 32), 0xA0, 0xf, 0xf, true);
            const double zodd = __longlong_as_double((long long)(((uint64_t)zh << 32) | zl));
            Y = act ? ((j & 1) ? zodd : z0) : 0.0;
            logA = row_bcast<1>(lg); // A is in the open interval (0,1)
        }

        const double q = row_sum<L>(Y * Y); // |Y|^2 for the Hastings term
        // The two exponentials of an iteration -- the size adaptation's multiplier and the Hastings term's exp(-q / 2) --
        // in ONE evaluation: odd lanes take the first argument, even lanes the second, lanes 1 and 0 of the row report.
        double e_adapt, e_hast;
        {
            const double xa = (ass != 0 ? a.d_cool_pow[min(li, a.cool_last)] : 0.0) * (rate - 0.234);
            const double ex = fm_exp_t((j & 1) ? xa : -0.5 * q, fm_tab<16>(FT_EXP, i));
            e_adapt = row_bcast<1>(ex);
            e_hast = row_bcast<0>(ex);
        }
        // ---- adaptation of covmat_proposal (adaptMCMC_source.cpp:117-139), replicated ------------------
        bool rescale = false, refactor = false;
        if (ass != 0 && i >= ass && (ashs == 0 || mul_rn(rate, di) < (double)ashs)) {
            // polynomial coefficients arrive by one scalar load per use (fastmath.h: fm_tab) instead of living in -- and
            // spilling from -- scalar registers across the loop; same operations, same bits as fm_exp / fm_log
            const double scaling_multiplier = e_adapt;
            scaling_sd = mul_rn(scaling_sd, scaling_multiplier);
            scaling_sd = (a.max_scaling_sd < scaling_sd) ? a.max_scaling_sd : scaling_sd;
            const double s2 = mul_rn(scaling_sd, scaling_sd);
            const bool any_small = mul_rn(s2, min_var) < 2E-16; // the smallest variance decides the test at :126
            rescale = !any_small && !dead;                     // covmat_proposal = scaling_sd^2 covmat (:129)
        } else if (ashs != 0 && mul_rn(rate, di) >= (double)ashs) {
            scaling_sd = a.shape_scaling_sd;
            refactor = !dead; // covmat_proposal = scaling_sd^2 covmat_empirical (:138)
            ++n_shape;
        }
        if (rescale) {
            if (base_emp) { // back from the shape phase: the factor of covmat again
#pragma unroll
                for (int k = 0; k < K; ++k) Rc[k] = R0c[k * L];
                prodR = prodR0;
                base_emp = false;
            }
            s_eff = scaling_sd;
        }
        // ---- R = chol(covmat_empirical) (:153): right-looking, in place in the R registers, one pivot at a time --
        // A divergent branch: the lanes of a row belong to one chain and take it together, so every cross-lane
        // operand comes from an active lane; the rows of chains that do not factor sit it out.
        if (refactor) {
            double (&Ac)[K] = Rc;
            if constexpr (CL) {
#pragma unroll
                for (int k = 0; k < K; ++k) Ac[k] = ((k >> 4) <= mj && mj < MB) ? Cl[16 * k] : 0.0;
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) Ac[k] = Cc[k];
            }
            bool ok = true;
            double prod = 1.0;
            [[maybe_unused]] const int half_base = 4 * ((int)(threadIdx.x & 63) & 32); // ds_bpermute address of lane 0 of this lane's chain (L = 32)
            int jo = j; // opaque: the 2 L lane masks j > p, j == p would otherwise be hoisted out of the iteration loop
            asm volatile("" : "+v"(jo)); // into scalar registers and come back one v_readlane at a time
            row_unroll<0, K>([&](auto pc) {
                constexpr int P = decltype(pc)::value;
                if (P < d) { // uniform
                    double piv;
                    if constexpr (CL) { // one chain per wavefront: the pivot is lane p's value, through a scalar register
                        const uint64_t b = (uint64_t)__double_as_longlong(Ac[P]);
                        const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)b, P), hi = __builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), P);
                        piv = __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
                    } else if constexpr (L == 32) {
                        // two chains per wavefront: the pivot of this lane's chain is lane (lane & 32) + P's value, fetched
                        // through the LDS crossbar (two ds_bpermute_b32 with the lane in the offset field): spread over
                        // both 16-lane rows and broadcast by DPP it was four moves, two row swaps, two DPP moves and three
                        // wait states per pivot, in a loop that is bound by instruction issue
                        // (asm: the builtin takes the whole address in a register, and the compiler then keeps one per pivot)
                        const uint64_t b = (uint64_t)__double_as_longlong(Ac[P]);
                        uint32_t lo, hi;
                        asm volatile("ds_bpermute_b32 %0, %2, %3 offset:%5\n\t"
                                     "ds_bpermute_b32 %1, %2, %4 offset:%5\n\t"
                                     "s_waitcnt lgkmcnt(0)"
                                     : "=&v"(lo), "=&v"(hi)
                                     : "v"(half_base), "v"((uint32_t)b), "v"((uint32_t)(b >> 32)), "i"(4 * P));
                        piv = __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
                    } else {
                        const RowVec<L> rowp(Ac[P]); // row p of the trailing matrix: entry j in lane j
                        piv = rowp.template get<P>();
                    }
                    ok = ok && (piv > 0.0);
                    double rjj, rinv;
                    fm_sqrt_rsq(piv, rjj, rinv);
                    prod *= rjj;
                    // finished row p of R: R_pj = a_pj / sqrt(a_pp) for j > p, the root itself at j = p, zero for j < p
                    // (lane p: a_pp / sqrt(a_pp), the root up to an ulp -- one lane mask instead of two)
                    const double rp = (jo >= P) ? Ac[P] * rinv : 0.0;
                    if constexpr (P + 1 < K) {
                        // a_ij -= R_pi R_pj for the rows i > p (only the part i <= j is ever read), in groups of four
                        // rows.  The group that holds p also meets rows i < p, where R_pi = 0 and nothing changes, and
                        // row p itself, which is overwritten with the finished row right after.
                        const RowVec<L> rv(-rp);
                        if constexpr (CL)
                            rv.template rows_from<P + 1, K>(Ac, rp, d);
                        else
                            rv.template rows_from<P + 1, K>(Ac, rp);
                    }
                    Ac[P] = rp;
                }
            });
            {   // the padding stays zero.  One compare against a per-lane bound, opaque to the compiler: the K loop-invariant
                // lane masks (k < d && act) it would otherwise hoist do not fit the scalar registers, and each came back
                // as two v_readlane per use (160 instructions per factorization at L = 32)
                int dj = act ? d : 0;
                asm volatile("" : "+v"(dj));
#pragma unroll
                for (int k = 0; k < K; ++k) Rc[k] = (k < dj) ? Ac[k] : 0.0;
            }
            prodR = prod;
            base_emp = true;
            s_eff = scaling_sd;
            if (!ok) { // arma::chol throws here; the chain stops and reports where
                dead = true;
                status = (i << 4) | MCMC_CHAIN_CHOL_FAILED;
                theta = __builtin_nan("");
            }
        }
        if (rescale || refactor) { // 1 / sqrt((2 pi)^d det(covmat_proposal)) (:56)
            double sp = 1.0, b = s_eff;
            for (int e = d; e > 0; e >>= 1) { // s_eff^d by repeated squaring
                if (e & 1) sp *= b;
                b *= b;
            }
            const double dt = sp * prodR;
            double det = dt * dt;
            if (!(det > 1e-280 && det < 1e280)) {
                // near under / overflow: the running product of the LU diagonal s^2 R_kk^2, factor by factor, as
                // arma::det forms it (see rwm_block.hip: when it reaches 0 decides when the reference starts to
                // reject everything through the NaN of :169-175)
                double rd = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) rd = (k == j) ? Rc[k] : rd;
                const RowVec<L> dv(rd);
                const double s2 = mul_rn(s_eff, s_eff);
                det = 1.0;
                row_unroll<0, K>([&](auto kc) {
                    constexpr int KK = decltype(kc)::value;
                    if (KK < d) {
                        const double rkk = dv.template get<KK>();
                        det = mul_rn(det, mul_rn(s2, mul_rn(rkk, rkk)));
                    }
                });
            }
            fm_sqrt_rsq(a.twopi_pow_d * det, sq_unused, rootTerm);
        }

        // ---- proposal theta + s (Y R) (:153-154) and |Y|^2 for the Hastings term --------------------------------
        const RowVec<L> yv(Y);
        double yr = 0.0;
        row_unroll<0, K / 8>([&](auto gc) {
            constexpr int G = decltype(gc)::value;
            yv.template dot8<G>(yr, Rc + 8 * G); // blocks past d multiply zeros: skipping them (uniform branches) measured slower
        });
        const double prop = theta + s_eff * yr;

        // ---- target and Metropolis-Hastings log ratio (:157-171) ------------------------------------------------
        const double t_prop = target(prop);
        double la;
        if (!isfinite(t_prop)) {
            la = -__builtin_inf();
        } else {
            la = t_prop - t_cur;
            const double pe = rootTerm * e_hast;
            double p = fm_log_t(pe, fm_tab<16>(FT_LOG, li));
            p = (pe == 0.0) ? -__builtin_inf() : p; // the special cases of fm_log<true>: the NaN-reject path needs them
            p = (pe < 0.0 || pe != pe) ? __builtin_nan("") : p;
            p = (pe == __builtin_inf()) ? pe : p;
            la = add_rn(la, p);
            la = add_rn(la, -p);
        }
        const bool accept = logA < la;
        if (accept) {
            theta = prop;
            t_cur = t_prop;
        }

        // ---- trace (:185): through the LDS tile, whole runs of consecutive chains per coordinate -----------------
        if (i == next_keep) { // uniform over the workgroup
            if (a.d_trace && a.trace_chain_major) { // [chain][t][k]: lane j writes coordinate j, d contiguous doubles per chain
                if (live && act) a.d_trace[((int64_t)c * a.n_kept + t_kept) * d + j] = theta;
            } else if (a.d_trace) {
                double *const tile = stage + (t_kept & 1) * (d * CPW);
                if (act) tile[j * CPW + cl] = theta;
                lds_barrier();
                double *const row = a.d_trace + t_kept * d * n + c0;
                for (int e = tid; e < d * CPW; e += NTHR) {
                    const int k = e / CPW, q2 = e - k * CPW;
                    if (c0 + q2 < n) row[(int64_t)k * n + q2] = tile[e];
                }
            }
            next_keep += a.thin;
            ++t_kept;
        }

        // ---- acceptance rate (:188-207), replicated ------------------------------------------------------------
        const double acc = accept ? 1.0 : 0.0;
        if (i == 1) {
            rate = acc;
        } else if (wgt == 0) {
            if (window == 0)
                rate = add_rn(rate, div_rn(add_rn(acc, -rate), di));
            else
                rate = accept_window_push(win, accept, window, win_pos, ring, n, live && j == 0);
        } else {
            rate = add_rn(mul_rn(rate, 1 - wgt), mul_rn(acc, wgt));
        }

        // ---- running mean / covariance (:210-214, update_sigma :42-45) ----------------------------------------
        if (need_moments) {
            const double inv_i = fm_rcp(di);
            const double sfac = (di - 1.0) * inv_i; // C (i-1)/i + ((i-1)/i^2 r_k) r_j
            const double res = act ? theta - mean : 0.0;
            mean += res * inv_i;
            const RowVec<L> bv((sfac * inv_i) * res);
            if constexpr (CL) {
                row_unroll<0, K / 16>([&](auto bc) {
                    c_block(bc, [&](double(&t)[16]) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) t[k] *= sfac;
                        bv.template rows16<decltype(bc)::value>(t, res);
                    });
                });
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) Cc[k] *= sfac;
                bv.template rows<K>(Cc, res);
            }
        }
    }

    if (live && act) {
        if (a.d_final_theta) a.d_final_theta[(int64_t)j * n + c] = theta;
        if (a.d_final_mean) a.d_final_mean[(int64_t)j * n + c] = mean;
        if (a.d_final_cov) { // the upper triangle, mirrored: the entries k > j of a lane's column are the same numbers up to
                             // rounding ((f r_k) r_j against (f r_j) r_k) and are never read -- the factorization uses k <= j
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (k <= j) {
                    const double cv = CL ? Cl[16 * k] : Cc[CL ? 0 : k];
                    a.d_final_cov[(int64_t)(j * d + k) * n + c] = cv;
                    a.d_final_cov[(int64_t)(k * d + j) * n + c] = cv;
                }
            }
        }
    }
    if (live && j == 0) {
        if (a.d_acceptance_rate) a.d_acceptance_rate[c] = rate;
        if (a.d_status) a.d_status[c] = status;
        if (a.d_shape_iterations) a.d_shape_iterations[c] = n_shape;
    }
    if (STATE && a.d_state_out && live) {
        double *const sp = a.d_state_out + c;
        if (act) {
            sp[(int64_t)(SL.THETA + j) * n] = theta;
            sp[(int64_t)(SL.MEAN + j) * n] = mean;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (k <= j) {
                    sp[(int64_t)(SL.C + tri(k, j)) * n] = CL ? Cl[16 * k] : Cc[CL ? 0 : k];
                    sp[(int64_t)(SL.R + tri(k, j)) * n] = Rc[k];
                }
            }
        }
        if (j == 0) {
            double *const ss = sp + (int64_t)SL.S * n;
            ss[ST_TAG * n] = (double)MCMC_KERNEL_ROW;
            ss[ST_ITER * n] = (double)(i0 + a.n_iterations);
            ss[ST_TCUR * n] = t_cur;
            ss[ST_RATE * n] = rate;
            ss[ST_SD * n] = scaling_sd;
            ss[ST_MULT * n] = 1.0;
            ss[ST_SEFF * n] = s_eff;
            ss[ST_PRODR * n] = prodR;
            ss[ST_ROOT * n] = rootTerm;
            ss[ST_WLO * n] = (double)(uint32_t)win.bits;
            ss[ST_WHI * n] = (double)(uint32_t)(win.bits >> 32);
            ss[ST_WLEN * n] = (double)win.len;
            ss[ST_BASE * n] = base_emp ? 1.0 : 0.0;
            ss[ST_STATUS * n] = (double)status;
            ss[ST_NSHAPE * n] = (double)n_shape;
            ss[ST_WCOUNT * n] = (double)win.count;
            for (int wd = 0; wd < a.win_words; ++wd) {
                const uint64_t v = (wd == (win_pos >> 6)) ? win.bits
                                                          : (uint64_t)__hip_atomic_load(ring + (int64_t)wd * n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sp[(int64_t)(SL.TOTAL + 2 * wd) * n] = (double)(uint32_t)v;
                sp[(int64_t)(SL.TOTAL + 2 * wd + 1) * n] = (double)(uint32_t)(v >> 32);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
bool rwm_row_serves(const RwmArgs &a, int target_kind)
{
    const bool full = a.d_Z || a.d_sigma_trace || a.d_accepted || a.d_log_acceptance || a.d_log_A || a.d_scaling_sd ||
                      a.d_scaling_multiplier || a.d_rate_trace;
    return !full && !a.factor_rank1 && target_kind == MCMC_TARGET_GAUSSIAN_PREC && a.d >= 1 && a.d <= kMaxRowDimRwm;
}

template <int L, int K>
static int launch_rwm_row_l(const RwmArgs &a, const RwmParams &p, hipStream_t s)
{
    using Lay = RowLayout<L>;
    const int d = a.d;
    std::vector<double> blob(Lay::TOTAL, 0.0);
    double min_var = p.cov_packed[0], prod0 = 1.0;
    const double *Hm = p.tparams.data() + d; // column-major precision matrix
    for (int k = 0; k < d; ++k) {
        min_var = std::fmin(min_var, p.cov_packed[tri(k, k)]);
        prod0 *= p.R0_packed[tri(k, k)];
        blob[Lay::TMEAN + k] = p.tparams[k];
        blob[Lay::INIT + k] = p.init[k];
        for (int jj = 0; jj < d; ++jj) {
            blob[Lay::HS + k * L + jj] = 0.5 * (Hm[(size_t)k * d + jj] + Hm[(size_t)jj * d + k]);
            if (k <= jj) blob[Lay::R0 + k * L + jj] = p.R0_packed[tri(k, jj)];
        }
    }
    blob[Lay::MINVAR] = min_var;
    blob[Lay::PROD0] = prod0;
    double *d_prm = nullptr;
    MCMC_HIP_CHECK(hipMallocAsync((void **)&d_prm, blob.size() * sizeof(double), s));
    hipError_t e = hipSuccess;
    const int ru = upload_block(d_prm, blob.data(), blob.size(), s);
    if (ru != MCMC_OK) {
        (void)hipFreeAsync(d_prm, s);
        return ru;
    }
    {
        // L = 64: as many chains (wavefronts) per workgroup as one CU's 160 KB of LDS holds, eight at most
        int cpw = Lay::CPW;
        if (L == 64) {
            cpw = 8;
            while (cpw > 1 && (size_t)Lay::lds_doubles(d, cpw) * sizeof(double) > 160 * 1024 - 512) --cpw;
        }
        const size_t ldsb = (size_t)Lay::lds_doubles(d, cpw) * sizeof(double);
        const unsigned grid = (unsigned)((a.n_chains + cpw - 1) / cpw), block = L == 64 ? 64u * cpw : 256u;
        auto go = [&](auto kern) {
            hipError_t e0 = ensure_dynamic_lds((const void *)kern, ldsb);
            if (e0 != hipSuccess) return e0;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(block), ldsb, s, a, (const double *)d_prm);
            return hipGetLastError();
        };
        e = (a.d_state_in || a.d_state_out) ? go(rwm_row_kernel<L, K, true>) : go(rwm_row_kernel<L, K, false>);
    }
    hipError_t e2 = hipFreeAsync(d_prm, s);
    if (e != hipSuccess) return set_error(MCMC_E_HIP, "rwm_row_kernel launch failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return set_error(MCMC_E_HIP, "hipFreeAsync failed: %s", hipGetErrorString(e2));
    return MCMC_OK;
}

int launch_rwm_row(const RwmArgs &a, const RwmParams &p, hipStream_t s)
{
    if (!rwm_row_serves(a, p.target_kind))
        return set_error(MCMC_E_UNSUPPORTED, "row kernel: Gaussian target, Philox streams, no per-iteration diagnostics, d <= %d", kMaxRowDimRwm);
    if ((int)p.tparams.size() != a.d + a.d * a.d) return set_error(MCMC_E_BAD_ARG, "gaussian target needs d + d*d parameters");
    if (a.d <= 16) return launch_rwm_row_l<16, 16>(a, p, s);
    if (a.d <= 24) return launch_rwm_row_l<32, 24>(a, p, s);
    if (a.d <= 32) return launch_rwm_row_l<32, 32>(a, p, s);
    if (a.d <= 48) return launch_rwm_row_l<64, 48>(a, p, s);
    return launch_rwm_row_l<64, 64>(a, p, s);
}

} // namespace mcmc
