// Adaptive random-walk Metropolis, one chain per lane, a SAMPLER and a GENERATOR wavefront per 64
// chains (gfx950), d <= 10.  Same loop as rwm_lane.hip (adaptMCMC_source.cpp:108-216; rwMCMC :252-313).
//
// Why two wavefronts.  The one-wave kernel needs > 256 registers, so one wave per SIMD is resident, and
// ONE wave issues an FP64 instruction only every ~11 cycles (tools/ubench_isa.hip, measured: 4.8 cycles
// is the SIMD's rate with several waves).  BASELINE config 2 has 65 536 chains = 1 024 waves for the
// 1 024 SIMDs of an MI355X, so more chains per SIMD are not available; the parallelism that is, is
// inside an iteration: everything that depends only on (seed, chain, iteration) -- the d Box-Muller
// normals from Philox, exp(-|Y|^2/2) of the Hastings term (:62), log(A) of the acceptance uniform
// (:174-175), the (i-1)/i and 1/i factors of update_sigma (:43) -- is 40 % of the instructions and has
// no dependence on the chain state.
//
// Layout.  A workgroup is 8 wavefronts = 256 chains: waves 0..3 sample chain groups 0..3, wave 4+g is
// the generator of group g.  On MI355X wave w and wave w+4 of a workgroup are placed on the same SIMD
// (tools/probe_hwid.hip, measured on all workgroups), so every SIMD holds one sampler and its own
// generator, whose instructions fill the sampler's issue gaps.  One workgroup per CU (LDS 157 KB at
// d = 10).  Per group, in LDS as [element][lane]: the draws of one iteration (generator -> sampler),
// theta of the finished iteration (sampler -> generator) and covmat_empirical.
//
// Division of work.  The generator draws one iteration ahead, and it also owns update_sigma (:42-45) and
// theta_mean (:212): it takes theta_i from the sampler and updates covmat_empirical in LDS while the
// sampler is already in iteration i+1.  The sampler reads covmat_empirical only in iterations in which
// one of its lanes factors it (shape phase), after waiting for the generator's "C_i complete" counter.
// The two waves synchronise through monotone counters in LDS (no workgroup barrier: the four pairs of a
// workgroup need different numbers of hand-shakes per iteration); every wait is bounded.
//
// What is done differently from the reference, results equal up to rounding:
//   * chol(s^2 Sigma) = s chol(Sigma): the kernel keeps the UNSCALED factor R (chol(covmat) in the size
//     phase, chol(covmat_empirical) in the shape phase) and the scale s apart, so the size phase needs
//     no factorization at all (the reference refactors s^2 covmat every iteration, :129,153) and the
//     shape phase factors covmat_empirical itself; the proposal is theta + s (Y R);
//   * update_sigma (:43) as C (i-1)/i + ((i-1)/i^2 r_a) r_b, one FMA per element; the loop is rotated so
//     that the adaptation and factorization for iteration i+1 close iteration i;
//   * mvrnorm_pdf (:48-64) from det = (s^d prod diag R)^2 and v^T Sigma^-1 v = |Y|^2, still evaluated
//     as log(rootTerm exp(-q/2)), added and subtracted, so the (la + p) - p rounding and the NaN
//     reject path survive;
//   * the Gaussian target -0.5 v^T H v is summed over the upper triangle of H (55 instead of 100 terms).
// The acceptance-rate recurrences (:188-207) and the phase tests (:117,131) use the reference's
// exact operation sequence (no FMA contraction): they feed discrete decisions.
#include "common.h"
#include "philox.h"
#include "targets.h"

namespace mcmc {

// target parameters as the kernel wants them
template <int KIND, int D>
struct PackedTarget;

// Gaussian by precision: m[D], diag(H)[D], then for j = 1..D-1, i < j: H_ij + H_ji at D + D + j(j-1)/2 + i
template <int D>
struct PackedTarget<MCMC_TARGET_GAUSSIAN_PREC, D> {
    static constexpr int NP = 2 * D + D * (D - 1) / 2;
#ifndef __HIPCC_RTC__
    static void pack(const std::vector<double> &tp, double *out)
    {
        for (int k = 0; k < D; ++k) out[k] = tp[k];
        const double *H = tp.data() + D; // column-major
        for (int j = 0; j < D; ++j) {
            out[D + j] = H[j * D + j];
            for (int i = 0; i < j; ++i) out[2 * D + j * (j - 1) / 2 + i] = H[j * D + i] + H[i * D + j];
        }
    }
#endif
    // The sampler keeps the chain state CENTRED, v = theta - m (one subtraction at the start, one addition
    // per stored value), so the target reads the state as it is: -0.5 sum_j v_j (H_jj v_j + sum_{i<j} (H_ij + H_ji) v_i)
    static constexpr bool CENTRED = true;
    // Rows J0 .. J1-1 form a group of at most kGroup coefficients (one row at least).  The coefficients of a group -- wave-
    // uniform LDS reads -- are all requested before the first is used, and scheduling fences keep it that way: left alone
    // the compiler, depending on the register budget of the variant, reads one pair, waits, uses it (25 exposed LDS round
    // trips per evaluation in the variant that has registers to spare; measured: 45.6 against 42.6 ms on C2).
    static constexpr int kGroup = 22;
    static constexpr int group_end(int j0)
    {
        int j1 = j0 + 1, cnt = j0 + 1;
        while (j1 < D && cnt + j1 + 1 <= kGroup) {
            cnt += j1 + 1;
            ++j1;
        }
        return j1;
    }
    template <int J0>
    __device__ static __forceinline__ void eval_rows(const double (&v)[D], const double *p, double &q)
    {
        constexpr int J1 = group_end(J0);
        constexpr int NO = J1 * (J1 - 1) / 2 - J0 * (J0 - 1) / 2; // off-diagonal coefficients of the group: contiguous in p
        double cd[J1 - J0], co[NO > 0 ? NO : 1];
#pragma unroll
        for (int j = J0; j < J1; ++j) cd[j - J0] = p[D + j];
#pragma unroll
        for (int e = 0; e < NO; ++e) co[e] = p[2 * D + J0 * (J0 - 1) / 2 + e];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            double g = cd[j - J0] * v[j];
#pragma unroll
            for (int i = 0; i < j; ++i) g += co[j * (j - 1) / 2 - J0 * (J0 - 1) / 2 + i] * v[i];
            q += v[j] * g;
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (J1 < D) eval_rows<J1>(v, p, q);
    }
    __device__ static __forceinline__ double eval(const double (&v)[D], const double *p)
    {
        double q = 0.0;
        eval_rows<0>(v, p, q);
        return -0.5 * q;
    }
};

struct PackedPrmView {
    const double *p;
    __device__ __forceinline__ double operator[](int i) const { return p[i]; }
};

template <>
struct PackedTarget<MCMC_TARGET_GOLDSTEIN_PRICE, 2> {
    static constexpr int NP = 0;
    static constexpr bool CENTRED = false;
#ifndef __HIPCC_RTC__
    static void pack(const std::vector<double> &, double *) {}
#endif
    __device__ static __forceinline__ double eval(const double (&x)[2], const double *p)
    {
        return Target<MCMC_TARGET_GOLDSTEIN_PRICE, 2>::eval(x, PackedPrmView{p});
    }
};

template <>
struct PackedTarget<MCMC_TARGET_BANANA, 2> {
    static constexpr int NP = 1;
    static constexpr bool CENTRED = false;
#ifndef __HIPCC_RTC__
    static void pack(const std::vector<double> &tp, double *out) { out[0] = tp[0]; }
#endif
    __device__ static __forceinline__ double eval(const double (&x)[2], const double *p)
    {
        return Target<MCMC_TARGET_BANANA, 2>::eval(x, PackedPrmView{p});
    }
};

#ifdef MCMC_USER_D
// a run-time compiled user target reads its parameters as the caller gave them
template <>
struct PackedTarget<MCMC_TARGET_USER, MCMC_USER_D> {
    static constexpr int NP = MCMC_USER_NP;
    static constexpr bool CENTRED = false;
    __device__ static __forceinline__ double eval(const double (&x)[MCMC_USER_D], const double *p)
    {
        return ::mcmc_user_target(x, p);
    }
};
#endif

// parameter block and LDS layout: pair_layout() in common.h
template <int D, int KIND, bool MOMENTS>
struct PairLayout {
    static constexpr PairLayoutRT L = pair_layout(D, PackedTarget<KIND, D>::NP, MOMENTS);
    static constexpr int T = L.T, NP = L.NP;
    static constexpr int COVD = L.COVD, TPRM = L.TPRM, SH = L.SH, INIT = L.INIT, R0 = L.R0, PROD0 = L.PROD0, TOTAL = L.TOTAL;
    static constexpr int DRAW = L.DRAW, BACK = L.BACK, CX = L.CX, GROUP = L.GROUP, LDS_DOUBLES = L.LDS_DOUBLES;
};

// a value that is the same in every lane, moved to a scalar register pair
__device__ __forceinline__ double wave_uniform(double x)
{
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// x^D by repeated squaring (D is a compile-time constant)
template <int D>
__device__ __forceinline__ double powi(double x)
{
    if constexpr (D == 1) {
        return x;
    } else if constexpr (D % 2 == 0) {
        const double h = powi<D / 2>(x);
        return h * h;
    } else {
        return x * powi<D - 1>(x);
    }
}

// ---- hand-shake between the two wavefronts of a chain group: monotone counters in LDS -----------
// Data is written with ordinary LDS stores, then the counter; LDS operations of one wave complete in
// order, so a reader that sees the counter sees the data.  The compiler fences keep it from moving LDS
// accesses across the counter operations.  Every wait is bounded: if a partner never arrives (a bug, not
// a data-dependent event) the wave gives up, raises the group's abort flag and the chain status says so.  A poll is
// ONE LDS round trip (the abort flag is looked at every 64th time) followed by a sleep of 8 x 64 cycles: the wave a
// poller waits for is the other wave of its own SIMD, and every poll takes issue slots and LDS cycles from it.  Measured
// on C2 (one box, two runs each): sleep 0 / 1 / 2 / 4 / 8 / 12 / 16 -> 38.9 / 38.4 / 38.3 / 38.0 / 38.0 / 37.9 / 38.6 ms;
// with two round trips per poll and sleep 1, as it was, 38.7.
constexpr int kSpinLimit = 1 << 21; // polls (~0.5 s) before a wave gives up
template <int V>
struct IntC {
    static constexpr int value = V;
};
// covmat_empirical goes from the generator to the sampler in kPairParts parts of whole columns (see update_part); part p
// is the columns [pair_part_col(d, p), pair_part_col(d, p + 1)).  d >= 6: the parts end at the first columns that lie
// behind a sixth and behind half of the T elements (d = 10: columns 4 and 7, i.e. 10 + 18 + 27 elements: the early parts
// are small because the sampler has little to do per column there); below that one part.
constexpr int kPairParts = 3;
__host__ __device__ constexpr int pair_part_col(int d, int part)
{
    if (part <= 0) return 0;
    if (part >= kPairParts || d < 6) return d;
    const int den = part == 1 ? 6 : 2;
    int cl = 0;
    while (den * tri(0, cl) < tri(0, d)) ++cl;
    return cl;
}

enum PairFlag { F_DRAW = 0, F_READ = 1, F_THETA = 2, F_BACKREAD = 3, F_C = 4, F_ABORT = 5 };

__device__ __forceinline__ void flag_set(int *flags, int which, int value)
{
    asm volatile("" ::: "memory");
    __hip_atomic_store(flags + which, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
// the counter as a wave-uniform (scalar) value, so that the polling loop is a scalar branch
__device__ __forceinline__ int flag_peek(int *flags, int which)
{
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(flags + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ bool flag_wait(int *flags, int which, int value)
{
    asm volatile("" ::: "memory");
    bool ok = true;
    int spin = 0;
#pragma nounroll
    while (flag_peek(flags, which) < value) {
        if (++spin > kSpinLimit || ((spin & 63) == 0 && flag_peek(flags, F_ABORT) != 0)) {
            ok = false;
            flag_set(flags, F_ABORT, 1);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    asm volatile("" ::: "memory");
    return ok;
}

// STATE adds checkpoint / resume (d_state_in, d_state_out, iter_offset); it is a variant of its own because
// the extra live values cost the C2 production kernel 3 % (measured).
// WIN: the acceptance rate is the mean of the last acceptance_window (<= 64) outcomes (:194-203).  A variant of its own
// because the window word, live but unused in the other two modes, was spilled, and its reload -- a vector-memory
// operation behind the ten trace stores -- made every iteration wait for those stores.
template <int D, int KIND, bool MOMENTS, bool STATE, bool WIN>
__global__ __launch_bounds__(128 * kPairGroups) void rwm_pair_kernel(const RwmArgs a, const double *__restrict__ prm)
{
    using Lay = PairLayout<D, KIND, MOMENTS>;
    constexpr int T = Lay::T;
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool generator = wave >= kPairGroups;
    const int group = wave & (kPairGroups - 1);
    const int64_t c = ((int64_t)blockIdx.x * kPairGroups + group) * 64 + lane;
    const int64_t n = a.n_chains;
    const bool live = c < n;
    const int64_t cc = live ? c : n - 1; // lanes past the end shadow the last chain and store nothing
    const uint64_t gchain = (uint64_t)(a.chain_offset + cc);
    const int n_iter = a.n_iterations;
    // checkpoint / resume: this launch runs iterations i0+1 .. i0+n_iter.  The hand-shake counters, the
    // cooling table and the trace rows count positions inside the launch (li); Philox counters, the
    // 1/i factors and the phase tests use the absolute iteration i0 + li.
    const int i0 = STATE ? a.iter_offset : 0;
    const double *const state_in = STATE ? a.d_state_in : nullptr;
    double *const state_out = STATE ? a.d_state_out : nullptr;
    constexpr RwmStateLayout SL = rwm_state_layout(D);

    for (int e = tid; e < Lay::SH; e += 128 * kPairGroups) lds[e] = prm[e];
    double *const G0 = lds + Lay::SH + group * Lay::GROUP;
    double *const Dl = G0 + lane;                                // draws: slot e at Dl[e * 64]
    double *const Bl = G0 + Lay::DRAW * 64 + lane;               // theta handed back: Bl[k * 64]
    double *const Xl = G0 + (Lay::DRAW + Lay::BACK) * 64 + lane; // covmat_empirical on request: Xl[e * 64]
    int *const flags = (int *)(G0 + (Lay::DRAW + Lay::BACK + Lay::CX) * 64);
    if (lane < 16 && !generator) flags[lane] = 0;
    if (MOMENTS && generator) { // covmat_empirical: zero, or as the checkpoint holds it (before the barrier:
                                // a resumed sampler may factor it in its prologue)
#pragma unroll
        for (int e = 0; e < T; ++e) Xl[e * 64] = state_in ? state_in[(int64_t)(SL.C + e) * n + cc] : 0.0;
    }
    __syncthreads();
    const double *const sh = lds;
    const double *const tp = sh + Lay::TPRM;
    constexpr bool CEN = PackedTarget<KIND, D>::CENTRED;
    auto shift = [&](int k) -> double { return CEN ? tp[k] : 0.0; }; // stored value = state + shift

    // ================================ generator ====================================================
    // Iteration i: the d normals, exp(-|Y|^2/2) and log A of iteration i (while the sampler works on
    // iteration i-1), then update_sigma / theta_mean for iteration i-1 from the theta the sampler handed
    // back.  theta_mean lives in THIS wave's registers, covmat_empirical in LDS as [element][lane] (Xl): this wave
    // updates it there, and the sampler reads it in iterations in which one of its lanes factors it (shape phase).
    if (generator) {
        const PhiloxKey key(a.seed);
        double mean[MOMENTS ? D : 1];
        if (MOMENTS) {
#pragma unroll
            for (int k = 0; k < D; ++k)
                mean[k] = state_in ? state_in[(int64_t)(SL.MEAN + k) * n + cc]
                                       : (a.d_init ? a.d_init[(int64_t)k * n + cc] : prm[Lay::INIT + k]) - shift(k);
        }
        bool ok = true;
        for (int i = 1; i <= n_iter + 1 && ok; ++i) { // i: position inside this launch
            double Y[D], eq = 0.0, logA = 0.0;
            if (i <= n_iter) {
                // log and sin/cos coefficients: one scalar load per iteration serves all Box-Muller pairs and log A
                const ScalarCoef<16> tl = fm_tab<16>(FT_LOG, i), ts = fm_tab<16>(FT_SINCOS, i);
#pragma unroll
                for (int b = 0; 2 * b < D; ++b) {
                    double z0, z1;
                    philox_normal2_t(key, gchain, (uint32_t)(i0 + i), TAG_RWM_NORMAL, (uint32_t)b, z0, z1, tl, ts);
                    Y[2 * b] = z0;
                    if (2 * b + 1 < D) Y[2 * b + 1] = z1;
                }
                double q = 0.0;
#pragma unroll
                for (int j = 0; j < D; ++j) q += Y[j] * Y[j];
                double A, u1;
                philox_uniform2(key, gchain, (uint32_t)(i0 + i), TAG_RWM_UNIFORM, 0u, A, u1);
                logA = fm_log_t(A, tl); // A is in the open interval (0,1)
                eq = fm_exp_t(-0.5 * q, fm_tab<16>(FT_EXP, i));
            }
            if (i <= n_iter) {
                if (i > 1) ok = flag_wait(flags, F_READ, i - 1); // the draw buffer is free again
#pragma unroll
                for (int k = 0; k < D; ++k) Dl[k * 64] = Y[k];
                Dl[D * 64] = eq;
                Dl[(D + 1) * 64] = logA;
                flag_set(flags, F_DRAW, i);
            }
            // theta of iteration i-1, as soon as the sampler has it
            double th[MOMENTS ? D : 1];
            if (MOMENTS && i > 1) {
                ok = ok && flag_wait(flags, F_THETA, i - 1);
#pragma unroll
                for (int k = 0; k < D; ++k) th[k] = Bl[k * 64];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // theta is in registers
                flag_set(flags, F_BACKREAD, i - 1);
            }
            if (MOMENTS && i > 1) {
                // update_sigma (:42-45) and the mean update (:212) of iteration i-1:
                // C (i-2)/(i-1) + ((i-2)/(i-1)^2 r_a) r_b, one FMA per element
                const double dm = (double)(i0 + i - 1);
                const double inv = fm_rcp(dm);
                const double sfac = (dm - 1.0) * inv, bfac = sfac * inv;
                double res[D];
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    res[k] = th[k] - mean[k];
                    mean[k] = mean[k] + res[k] * inv;
                }
                // in batches of up to kUB elements: all reads of a batch are requested before the first is used (left alone the
                // compiler keeps one read ahead and waits for it T / 2 times; the sampler waits for F_C meanwhile)
                constexpr int kUB = 14;
                double bc[D];
#pragma unroll
                for (int k = 0; k < D; ++k) bc[k] = bfac * res[k];
                // In parts of whole columns, each announced by itself (F_C = kPairParts (i-1) - (kPairParts-1) + part): the
                // sampler factors column by column and starts on the first part while this wave still updates the others
                // (waiting for all of covmat_empirical it sat out the whole update: ~240 instructions of ONE wave per iteration)
                auto update_part = [&](auto pc) {
                    constexpr int PART = decltype(pc)::value;
                    constexpr int EA = tri(0, pair_part_col(D, PART)), EB = tri(0, pair_part_col(D, PART + 1));
                    if constexpr (EA < EB) {
                        constexpr int NB = (EB - EA + kUB - 1) / kUB, UB = (EB - EA + NB - 1) / NB; // batches of equal size
#pragma unroll
                        for (int e0 = EA; e0 < EB; e0 += UB) {
                            double x[UB];
#pragma unroll
                            for (int u = 0; u < UB; ++u)
                                if (e0 + u < EB) x[u] = Xl[(e0 + u) * 64];
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int u = 0; u < UB; ++u) {
                                if (e0 + u < EB) {
                                    const int cl = tri_col(e0 + u), r = e0 + u - tri(0, cl); // constants once unrolled
                                    x[u] = fma(x[u], sfac, bc[cl] * res[r]);
                                    Xl[(e0 + u) * 64] = x[u];
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        // (no wait: the counter store follows the element stores of this wave through the same LDS queue)
                        flag_set(flags, F_C, kPairParts * (i - 1) - (kPairParts - 1) + PART); // these columns of covmat_empirical of iteration i-1 are complete
                    }
                };
                update_part(IntC<0>{});
                update_part(IntC<1>{});
                update_part(IntC<2>{});
                static_assert(kPairParts == 3, "one update_part call per part");
            }
        }
        if (MOMENTS && live) {
#pragma unroll
            for (int k = 0; k < D; ++k)
                if (a.d_final_mean) a.d_final_mean[(int64_t)k * n + c] = mean[k] + shift(k);
            if (a.d_final_cov) {
#pragma unroll
                for (int cl = 0; cl < D; ++cl) {
#pragma unroll
                    for (int r = 0; r < D; ++r)
                        a.d_final_cov[(int64_t)(cl * D + r) * n + c] = Xl[(r <= cl ? tri(r, cl) : tri(cl, r)) * 64];
                }
            }
            if (state_out) { // this wave's half of the checkpoint: theta_mean (as kept: centred) and covmat_empirical
#pragma unroll
                for (int k = 0; k < D; ++k) state_out[(int64_t)(SL.MEAN + k) * n + c] = mean[k];
#pragma unroll
                for (int e = 0; e < T; ++e) state_out[(int64_t)(SL.C + e) * n + c] = Xl[e * 64];
            }
        }
        return;
    }

    // ================================ sampler ======================================================
    const int ass = a.adapt_size_start, ashs = a.adapt_shape_start;
    const int window = a.acceptance_window;
    const double w = a.acceptance_rate_weight;
    const double min_var = wave_uniform(sh[Lay::COVD]); // smallest diagonal entry of covmat (host)
    double theta[D], R[T]; // theta: the chain state (minus the target mean if CENTRED); R: unscaled upper factor,
                           // R^T R = covmat (size phase) or covmat_empirical (shape phase)
#pragma unroll
    for (int k = 0; k < D; ++k) theta[k] = (a.d_init ? a.d_init[(int64_t)k * n + cc] : prm[Lay::INIT + k]) - shift(k);
#pragma unroll
    for (int e = 0; e < T; ++e) R[e] = prm[Lay::R0 + e];
    double prodR = prm[Lay::PROD0]; // product of diag(R)
    bool base_emp = false;          // R holds chol(covmat_empirical)
    double s_eff = 1.0;             // scale of covmat_proposal = s_eff^2 R^T R
    double rootTerm, sq_unused;
    fm_sqrt_rsq(a.twopi_pow_d * (prodR * prodR), sq_unused, rootTerm);
    double t_cur = PackedTarget<KIND, D>::eval(theta, tp);
    double rate = 0.0, scaling_sd = 1.0;
    [[maybe_unused]] uint64_t win_bits = 0;
    [[maybe_unused]] int win_len = 0;
    int32_t status = MCMC_CHAIN_OK, n_shape = 0;
    bool dead = false, sync_ok = true;
    if (state_in) { // resume: everything the loop carries from one iteration to the next
        const double *const sp = state_in + cc;
#pragma unroll
        for (int k = 0; k < D; ++k) theta[k] = sp[(int64_t)(SL.THETA + k) * n];
#pragma unroll
        for (int e = 0; e < T; ++e) R[e] = sp[(int64_t)(SL.R + e) * n];
        const double *const ss = sp + (int64_t)SL.S * n;
        t_cur = ss[ST_TCUR * n];
        rate = ss[ST_RATE * n];
        scaling_sd = ss[ST_SD * n];
        s_eff = ss[ST_SEFF * n];
        prodR = ss[ST_PRODR * n];
        rootTerm = ss[ST_ROOT * n];
        if constexpr (WIN) {
            win_bits = ((uint64_t)(uint32_t)ss[ST_WHI * n] << 32) | (uint64_t)(uint32_t)ss[ST_WLO * n];
            win_len = (int)ss[ST_WLEN * n];
        }
        base_emp = ss[ST_BASE * n] != 0.0;
        status = (int32_t)ss[ST_STATUS * n];
        n_shape = (int32_t)ss[ST_NSHAPE * n];
        dead = status != MCMC_CHAIN_OK;
        if (ss[ST_TAG * n] != (double)MCMC_KERNEL_PAIR || ss[ST_ITER * n] != (double)i0) { // not a state this launch can continue
            dead = true;
            status = ((i0 + 1) << 4) | MCMC_CHAIN_BAD_STATE;
#pragma unroll
            for (int k = 0; k < D; ++k) theta[k] = __builtin_nan("");
        }
    }
    int next_keep = (i0 / a.thin + 1) * a.thin; // next iteration whose state is stored
    int64_t t_kept = 0;

    // li = 0 is the prologue: the adaptation for iteration i0 + 1 only
    for (int li = 0; li <= n_iter && sync_ok; ++li) {
        const int i = i0 + li;
        // pow(adapt_size_cooling, (i+1) - adapt_size_start) for the adaptation at the end of this iteration:
        // the scalar load is issued here so that its latency is covered by the sampling step
        const double cool = (ass != 0 && li < n_iter) ? a.d_cool_pow[min(li + 1, a.cool_last)] : 0.0;
        if (li > 0) {
            double Y[D];
            sync_ok = flag_wait(flags, F_DRAW, li);
#pragma unroll
            for (int k = 0; k < D; ++k) Y[k] = Dl[k * 64];
            const double eq = Dl[D * 64];
            const double logA = Dl[(D + 1) * 64];
            __builtin_amdgcn_sched_barrier(0); // no arithmetic between the reads: it would wait for them one by one
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the draws are in registers
            flag_set(flags, F_READ, li);

            // ---- proposal theta + Y chol(covmat_proposal) = theta + s (Y R) (:149-154) -----------
            double prop[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k <= j; ++k) s += Y[k] * R[tri(k, j)];
                prop[j] = theta[j] + s_eff * s;
            }
            // ---- target and Metropolis-Hastings log ratio (:157-171) ----------------------------
            const double t_prop = PackedTarget<KIND, D>::eval(prop, tp);
            double la;
            if (!isfinite(t_prop)) {
                la = -__builtin_inf();
            } else {
                la = t_prop - t_cur;
                const double p = fm_log<true>(rootTerm * eq); // mvrnorm_pdf, both directions (:48-64)
                la = add_rn(la, p);
                la = add_rn(la, -p);
            }
            // ---- accept / reject (:174-182) ------------------------------------------------------
            const bool accept = logA < la;
            if (accept) {
#pragma unroll
                for (int k = 0; k < D; ++k) theta[k] = prop[k];
                t_cur = t_prop;
            }
            // ---- hand theta of iteration i to the generator (update_sigma / mean, :210-214) ---------
            // as soon as it is decided: the generator updates covmat_empirical while this wave stores the trace row and
            // adapts, and the factorization below waits for that update
            if (MOMENTS) {
                if (li > 1) sync_ok = flag_wait(flags, F_BACKREAD, li - 1) && sync_ok; // the generator has taken theta of iteration i-1
#pragma unroll
                for (int k = 0; k < D; ++k) Bl[k * 64] = theta[k];
                flag_set(flags, F_THETA, li);
            }
            // ---- trace (:185) -------------------------------------------------------------------
            if (i == next_keep) {
                if (a.d_trace && live) {
                    double *const row = a.d_trace + t_kept * D * n; // wave-uniform part of the address
                    // the D shifts (wave-uniform LDS reads) are requested together: read where they are used, each pair was
                    // a round trip of its own in front of its two stores
                    double sv[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) sv[k] = shift(k);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < D; ++k) row[(int64_t)k * n + c] = theta[k] + sv[k];
                }
                next_keep += a.thin;
                ++t_kept;
            }
            // ---- acceptance rate (:188-207): the reference's exact operation sequence ------------
            const double acc = accept ? 1.0 : 0.0;
            const double di = (double)i;
            if (i == 1) {
                rate = acc;
            } else if (w == 0) {
                if (!WIN || window == 0) { // the launcher takes the WIN = false variant whenever window == 0
                    rate = add_rn(rate, div_rn(add_rn(acc, -rate), di));
                } else {
                    win_bits = (win_bits << 1) | (accept ? 1ull : 0ull);
                    if (win_len < window) ++win_len;
                    const uint64_t m = (win_len >= 64) ? ~0ull : ((1ull << win_len) - 1ull);
                    rate = div_rn((double)__popcll(win_bits & m), (double)win_len);
                }
            } else {
                rate = add_rn(mul_rn(rate, 1 - w), mul_rn(acc, w));
            }
        }

        // ---- adaptation of covmat_proposal for iteration i + 1 (:117-139) -----------------------
        bool rescale = false, refactor = false;
        if (li < n_iter) {
            const int in = i + 1;
            const double din = (double)in;
            if (ass != 0 && in >= ass && (ashs == 0 || mul_rn(rate, din) < (double)ashs)) {
                const double scaling_multiplier = fm_exp(cool * (rate - 0.234));
                scaling_sd = mul_rn(scaling_sd, scaling_multiplier);
                scaling_sd = (a.max_scaling_sd < scaling_sd) ? a.max_scaling_sd : scaling_sd; // std::min
                const double s2 = mul_rn(scaling_sd, scaling_sd);
                // "any diagonal entry of the new matrix < 2e-16" (:126): the smallest variance decides (rounding is monotone)
                const bool any_small = mul_rn(s2, min_var) < 2E-16;
                rescale = !any_small && !dead; // covmat_proposal = scaling_sd^2 covmat (:129)
            } else if (ashs != 0 && mul_rn(rate, din) >= (double)ashs) {
                scaling_sd = a.shape_scaling_sd;
                refactor = MOMENTS && i > 0 && !dead; // covmat_proposal = scaling_sd^2 covmat_empirical (:138)
                ++n_shape;
            }
        }
        const bool want_c = MOMENTS && __builtin_amdgcn_ballot_w64(refactor) != 0;
        if (rescale) {
            if (base_emp) { // back from the shape phase: the factor of covmat again
#pragma unroll
                for (int e = 0; e < T; ++e) R[e] = prm[Lay::R0 + e];
                prodR = prm[Lay::PROD0];
                base_emp = false;
            }
            s_eff = scaling_sd;
        }
        // ---- R = chol(covmat_empirical) for iteration i + 1 (:153) ------------------------------
        if constexpr (MOMENTS) {
            if (want_c) {
                bool ok = true;
                double prod = 1.0;
                double rinv[D];
                // covmat_empirical is read INTO the registers of the factor it replaces, all elements of a part requested back
                // to back, and factored in place.  Read where it is used -- one element, wait, use, the next element -- the
                // factorization spent T / 2 exposed LDS round trips (~3 400 of the ~11 900 cycles of a shape-phase
                // iteration, measured) on 165 FMAs.  Lanes that do not refactor are masked off and keep their factor.
                // Column cl needs column cl of covmat_empirical and the finished columns before it: the first parts are
                // factored while the generator still updates the later ones (see update_part).
                auto factor_part = [&](auto pc) {
                    constexpr int PART = decltype(pc)::value;
                    constexpr int CA = pair_part_col(D, PART), CB = pair_part_col(D, PART + 1);
                    if constexpr (CA < CB) {
                        sync_ok = sync_ok && flag_wait(flags, F_C, kPairParts * li - (kPairParts - 1) + PART);
                        if (refactor) {
#pragma unroll
                            for (int e = tri(0, CA); e < tri(0, CB); ++e) R[e] = Xl[e * 64];
#pragma unroll
                            for (int cl = CA; cl < CB; ++cl) {
#pragma unroll
                                for (int r = 0; r < cl; ++r) {
                                    double s = R[tri(r, cl)];
#pragma unroll
                                    for (int m = 0; m < r; ++m) s -= R[tri(m, r)] * R[tri(m, cl)];
                                    R[tri(r, cl)] = s * rinv[r];
                                }
                                double ajj = R[tri(cl, cl)];
#pragma unroll
                                for (int m = 0; m < cl; ++m) ajj -= R[tri(m, cl)] * R[tri(m, cl)];
                                ok = ok && (ajj > 0.0);
                                double rjj;
                                fm_sqrt_rsq(ajj, rjj, rinv[cl]);
                                R[tri(cl, cl)] = rjj;
                                prod *= rjj;
                            }
                        }
                    }
                };
                factor_part(IntC<0>{});
                factor_part(IntC<1>{});
                factor_part(IntC<2>{});
                if (refactor) {
                    prodR = prod;
                    base_emp = true;
                    s_eff = scaling_sd;
                    if (!ok) { // arma::chol throws here; the chain stops and reports where
                        dead = true;
                        status = ((i + 1) << 4) | MCMC_CHAIN_CHOL_FAILED;
#pragma unroll
                        for (int k = 0; k < D; ++k) theta[k] = __builtin_nan("");
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // done with the copy before the next hand-back
            }
        }
        if (rescale || refactor) { // 1 / sqrt((2 pi)^d det(covmat_proposal)) (:56)
            const double dt = powi<D>(s_eff) * prodR;
            double sq;
            fm_sqrt_rsq(a.twopi_pow_d * (dt * dt), sq, rootTerm);
        }
    }

    if (live) {
#pragma unroll
        for (int k = 0; k < D; ++k)
            if (a.d_final_theta) a.d_final_theta[(int64_t)k * n + c] = theta[k] + shift(k);
        if (a.d_acceptance_rate) a.d_acceptance_rate[c] = rate;
        if (a.d_status) a.d_status[c] = sync_ok ? status : (MCMC_CHAIN_SYNC_LOST | ((i0 + n_iter) << 4));
        if (a.d_shape_iterations) a.d_shape_iterations[c] = n_shape;
        if (MOMENTS && state_out) { // this wave's half of the checkpoint (the generator writes theta_mean and covmat_empirical)
            double *const sp = state_out + c;
#pragma unroll
            for (int k = 0; k < D; ++k) sp[(int64_t)(SL.THETA + k) * n] = theta[k];
#pragma unroll
            for (int e = 0; e < T; ++e) sp[(int64_t)(SL.R + e) * n] = R[e];
            double *const ss = sp + (int64_t)SL.S * n;
            ss[ST_TAG * n] = (double)MCMC_KERNEL_PAIR;
            ss[ST_ITER * n] = (double)(i0 + n_iter);
            ss[ST_TCUR * n] = t_cur;
            ss[ST_RATE * n] = rate;
            ss[ST_SD * n] = scaling_sd;
            ss[ST_MULT * n] = 1.0;
            ss[ST_SEFF * n] = s_eff;
            ss[ST_PRODR * n] = prodR;
            ss[ST_ROOT * n] = rootTerm;
            ss[ST_WLO * n] = WIN ? (double)(uint32_t)win_bits : 0.0;
            ss[ST_WHI * n] = WIN ? (double)(uint32_t)(win_bits >> 32) : 0.0;
            ss[ST_WLEN * n] = WIN ? (double)win_len : 0.0;
            ss[ST_BASE * n] = base_emp ? 1.0 : 0.0;
            ss[ST_STATUS * n] = (double)(sync_ok ? status : (MCMC_CHAIN_SYNC_LOST | ((i0 + n_iter) << 4)));
            ss[ST_NSHAPE * n] = (double)n_shape;
            ss[ST_WCOUNT * n] = 0.0; // windows above 64 outcomes run in the one-wave kernel (rwm_pair_serves)
        }
    }
}

#ifndef __HIPCC_RTC__
// ------------------------------------------------------------------------------------------
int launch_rwm_pair_with(const RwmArgs &a, const RwmParams &p, hipStream_t s, int n_packed,
                         const std::vector<double> &packed_tparams, const RwmLauncher &launch)
{
    const int D = a.d;
    const bool state = a.d_state_in || a.d_state_out;
    const bool moments = a.adapt_shape_start != 0 || a.d_final_mean || a.d_final_cov || state;
    const PairLayoutRT L = pair_layout(D, n_packed, moments); // the HBM block does not depend on `moments`
    if ((int)packed_tparams.size() != n_packed) return set_error(MCMC_E_BAD_ARG, "pair kernel: target parameter block of the wrong size");
    std::vector<double> blob(L.TOTAL, 0.0);
    double prod0 = 1.0;
    double min_var = p.cov_packed[0];
    for (int k = 0; k < D; ++k) {
        min_var = std::fmin(min_var, p.cov_packed[tri(k, k)]);
        blob[L.INIT + k] = p.init[k];
        prod0 *= p.R0_packed[tri(k, k)];
    }
    blob[L.COVD] = min_var;
    for (int e = 0; e < n_packed; ++e) blob[L.TPRM + e] = packed_tparams[e];
    for (int e = 0; e < L.T; ++e) blob[L.R0 + e] = p.R0_packed[e];
    blob[L.PROD0] = prod0;
    double *d_prm = nullptr;
    MCMC_HIP_CHECK(hipMallocAsync((void **)&d_prm, blob.size() * sizeof(double), s));
    if (const int ru = upload_block(d_prm, blob.data(), blob.size(), s); ru != MCMC_OK) {
        (void)hipFreeAsync(d_prm, s);
        return ru;
    }
    const int64_t grid = (a.n_chains + 64 * kPairGroups - 1) / (64 * kPairGroups);
    hipError_t e1 = launch(state ? 2 : (moments ? 1 : 0), (unsigned)grid, 128u * kPairGroups, (size_t)L.LDS_DOUBLES * sizeof(double), s, a, d_prm);
    hipError_t e2 = hipFreeAsync(d_prm, s);
    if (e1 != hipSuccess) return set_error(MCMC_E_HIP, "rwm_pair_kernel launch failed: %s", hipGetErrorString(e1));
    if (e2 != hipSuccess) return set_error(MCMC_E_HIP, "hipFreeAsync failed: %s", hipGetErrorString(e2));
    return MCMC_OK;
}

template <int D, int KIND>
static int launch_rwm_pair_dk(const RwmArgs &a, const RwmParams &p, hipStream_t s)
{
    if ((int)p.tparams.size() != target_param_count(KIND, D))
        return set_error(MCMC_E_BAD_ARG, "target kind %d with d=%d needs %d parameters, got %d", KIND, D,
                         target_param_count(KIND, D), (int)p.tparams.size());
    std::vector<double> packed(PackedTarget<KIND, D>::NP, 0.0);
    PackedTarget<KIND, D>::pack(p.tparams, packed.data());
    return launch_rwm_pair_with(
        a, p, s, PackedTarget<KIND, D>::NP, packed,
        [](int variant, unsigned grid, unsigned block, size_t ldsb, hipStream_t st, const RwmArgs &aa, const double *d_prm) {
            auto go = [&](auto kern) {
                hipError_t e0 = ensure_dynamic_lds((const void *)kern, ldsb);
                if (e0 != hipSuccess) return e0;
                hipLaunchKernelGGL(kern, dim3(grid), dim3(block), ldsb, st, aa, d_prm);
                return hipGetLastError();
            };
            if (aa.acceptance_rate_weight == 0 && aa.acceptance_window != 0)
                return variant == 2 ? go(rwm_pair_kernel<D, KIND, true, true, true>)
                                    : (variant == 1 ? go(rwm_pair_kernel<D, KIND, true, false, true>)
                                                    : go(rwm_pair_kernel<D, KIND, false, false, true>));
            return variant == 2 ? go(rwm_pair_kernel<D, KIND, true, true, false>)
                                : (variant == 1 ? go(rwm_pair_kernel<D, KIND, true, false, false>)
                                                : go(rwm_pair_kernel<D, KIND, false, false, false>));
        });
}

bool rwm_pair_serves(const RwmArgs &a)
{
    const bool full = a.d_Z || a.d_sigma_trace || a.d_accepted || a.d_log_acceptance || a.d_log_A || a.d_scaling_sd ||
                      a.d_scaling_multiplier || a.d_rate_trace;
    // a window of more than 64 outcomes needs the ring of accept_window.h: two more live values in a sampler wave that
    // already spills; those runs go to the one-wave kernel
    return !full && a.d >= 1 && a.d <= kMaxPairDimRwm && a.win_words == 0;
}

int launch_rwm_pair(const RwmArgs &a, const RwmParams &p, hipStream_t s)
{
    if (!rwm_pair_serves(a))
        return set_error(MCMC_E_UNSUPPORTED, "paired-wave kernel: Philox streams, no per-iteration diagnostics, d <= %d, acceptance_window <= 64",
                         kMaxPairDimRwm);
    if (p.target_kind == MCMC_TARGET_USER) return launch_rwm_pair_user(a, p, s);
    if (p.target_kind == MCMC_TARGET_GOLDSTEIN_PRICE) {
        if (a.d != 2) return set_error(MCMC_E_BAD_ARG, "Goldstein-Price target is 2-dimensional (d=%d)", a.d);
        return launch_rwm_pair_dk<2, MCMC_TARGET_GOLDSTEIN_PRICE>(a, p, s);
    }
    if (p.target_kind == MCMC_TARGET_BANANA) {
        if (a.d != 2) return set_error(MCMC_E_BAD_ARG, "banana target is 2-dimensional (d=%d)", a.d);
        return launch_rwm_pair_dk<2, MCMC_TARGET_BANANA>(a, p, s);
    }
    if (p.target_kind != MCMC_TARGET_GAUSSIAN_PREC)
        return set_error(MCMC_E_BAD_ARG, "unknown target kind %d", p.target_kind);
    switch (a.d) {
#define MCMC_CASE(D) \
    case D: return launch_rwm_pair_dk<D, MCMC_TARGET_GAUSSIAN_PREC>(a, p, s);
        MCMC_CASE(1) MCMC_CASE(2) MCMC_CASE(3) MCMC_CASE(4) MCMC_CASE(5) MCMC_CASE(6)
        MCMC_CASE(7) MCMC_CASE(8) MCMC_CASE(9) MCMC_CASE(10)
#undef MCMC_CASE
    default: return set_error(MCMC_E_UNSUPPORTED, "paired-wave kernel: d=%d outside 1..%d", a.d, kMaxPairDimRwm);
    }
}

#endif // !__HIPCC_RTC__

} // namespace mcmc
