// Adaptive random-walk Metropolis for 12 < d <= 128: one chain per 256-thread workgroup (gfx950).
//
// Replaces the main loop of adaptMCMC / rwMCMC (adaptMCMC_source.cpp:108-216, :252-313) where the
// per-chain state no longer fits one lane: at d = 100 the packed Cholesky factor and the packed
// empirical covariance are 40 KB each.
//
// Data placement per chain (= per workgroup of four wavefronts, one per SIMD of a CU):
//   * R, the upper Cholesky factor of covmat_proposal, packed by columns, in LDS (T = d(d+1)/2
//     doubles) together with theta, theta_mean, the normals Y of four iterations, the proposal and two scratch
//     vectors: 51 KB at d = 100; the registers (two copies of the 28-tile matrix) allow two chains per CU there;
//   * covmat_empirical in REGISTERS, 2-D cyclic over the 16 x 16 threads: thread (ti, tk) owns the
//     elements (16a + ti, 16b + tk), a <= b, of the upper triangle (NT(NT+1)/2 doubles, NT = ceil(d/16)).
//     It is only ever touched element-wise (the rank-1 update_sigma), and the factorization works
//     on a register copy s^2 * C in the same layout, so no thread needs another's slice;
//   * shared parameters (covmat, chol(covmat), target mean and precision) stay in HBM/L2 and are
//     read coalesced (the precision matrix column-major, thread i reading row i).
// Scalars of the chain (acceptance rate, scaling_sd, cached target value, ...) are replicated in
// every thread, so decisions are taken without a broadcast.
//
// chol(s^2 Sigma) = s chol(Sigma): R in LDS is the UNSCALED factor -- chol(covmat) while the size phase
// runs, chol(covmat_empirical) in the shape phase -- and the scale s is kept apart (the proposal is
// theta + s (Y R)).  The size phase therefore needs no factorization (the reference refactors
// scaling_sd^2 covmat every iteration, adaptMCMC_source.cpp:129,153: d^3/3 flops for the same matrix).
//
// The factorization is the right-looking (outer-product) Cholesky with the trailing matrix held in
// registers: at step j the owners of row j (a quarter of ONE wave) publish it in a double-buffered
// LDS row -- unscaled, zero up to the pivot, the pivot in a slot of its own -- ONE barrier, then every
// thread reads the 2 NT entries it needs without any range test, scales one side by 1/pivot and
// updates its tile with at most NT(NT+1)/2 FMAs (a_ik - (a_ji / a_jj) a_jk).  No square root is taken
// inside the loop: the unscaled rows and the pivots are kept, and all d square roots and the scaling of
// the rows by 1 / R_jj are done once, in parallel, after the last pivot.
// The step is instruction-bound, so what counts is the number of instructions around the FMAs.
#ifndef __HIPCC_RTC__
#include <algorithm>
#endif

#include "accept_window.h"
#include "common.h"
#include "philox.h"
#include "targets.h"

// A user target for 12 < d <= 128 exists only inside a run-time compiled module (user_target.hip defines
// MCMC_USER_BLOCK ahead of this file and appends the user's definition): the log-density as a sum of `nparts`
// partial terms, one per thread of the chain's workgroup; x is the point (in LDS), the kernel adds the parts up.
#ifdef MCMC_USER_BLOCK
__device__ double mcmc_user_target_part(const double *x, int d, const double *params, int part, int nparts);
#endif

namespace mcmc {

constexpr int kBlockThreads = 256;
__host__ __device__ constexpr int blk(int a, int b) { return b * (b + 1) / 2 + a; } // tile block (a,b), a <= b

// parameter block (doubles): init[d] cov[T] R0[T] root0 tmean[d] Hdiag[d] meta[2] ttab[256][2] ptab[256][3] pred[d][2] Hq[TP][256]
//   cov, R0 = chol(covmat): packed by columns, (r,c) r<=c at c(c+1)/2 + r
//   The two triangular sums of an iteration -- the target's sum_c v_c sum_{i<c} (H_ic + H_ci) v_i and the proposal's
//   sum_{k<=c} Y_k R_kc -- are cut into runs of consecutive i (k) of one row (column), at most TP (SP) long, one run
//   per thread (block_segments): ttab[tid] = (row, first i), ptab[tid] = (column, first k, length), -1 = no run;
//   pred[c] = (first thread, number of threads) of column c.  Hq[t][tid] is the coefficient of term t of thread
//   tid's target run, zero beyond its end.  meta = (TP, KL): the first KL rows of Hq are copied into LDS by every
//   workgroup (as many as the CU's LDS holds beside the chains' own state), the rest is read from L2.
struct BlockLayout {
    int d, T;
    __host__ __device__ explicit BlockLayout(int d_) : d(d_), T(d_ * (d_ + 1) / 2) {}
    __host__ __device__ int init() const { return 0; }
    __host__ __device__ int cov() const { return d; }
    __host__ __device__ int r0() const { return d + T; }
    __host__ __device__ int root0() const { return d + 2 * T; }
    __host__ __device__ int tmean() const { return d + 2 * T + 1; }
    __host__ __device__ int hdiag() const { return 2 * d + 2 * T + 1; }
    __host__ __device__ int meta() const { return 3 * d + 2 * T + 1; }
    __host__ __device__ int ttab() const { return meta() + 2; }
    __host__ __device__ int ptab() const { return ttab() + 2 * kBlockThreads; }
    __host__ __device__ int pred() const { return ptab() + 3 * kBlockThreads; }
    __host__ __device__ int hq() const { return pred() + 2 * d; }
    static constexpr int kHqPad = 8; // zero rows behind Hq: the target sum loads its coefficients in batches of up to 8, unguarded
    __host__ __device__ int total(int TP) const { return hq() + (TP + kHqPad) * kBlockThreads; }
    // LDS: R, theta, mean, Y of four iterations, proposal, v + zero pad, scalars, pivot rows, diag products, per-thread partial sums
    __host__ __device__ int lds_doubles() const { return T + 8 * d + kVPad + 16 + 2 * (16 * ((d + 15) / 16) + 2) + 16 + kBlockThreads; }
    static constexpr int kVPad = 64; // zeros after v: a target run may read up to TP + 7 entries past its row's last term
    // MCMC_FACTOR_RANK1: six vectors of fx() doubles (residual, whitened residual, p, g, u, diag R; a proposal run may read
    // up to 31 entries past d, so every vector is padded and zero-filled) and one partial sum per thread
    __host__ __device__ int fx() const { return 16 * ((d + 15) / 16) + 32; }
    __host__ __device__ int fast_doubles() const { return 6 * fx() + kBlockThreads; }
};

// Length of the runs block_segments (below) cuts rows c = 0..d-1 of c + extra terms into: the smallest S >= min_run
// that needs no more than kBlockThreads runs.  constexpr: the kernels size their per-thread run registers with it.
__host__ __device__ constexpr int block_run_length(int d, int extra, int min_run)
{
    for (int S = min_run;; ++S) {
        long runs = 0;
        for (int c = 0; c < d; ++c) runs += (c + extra + S - 1) / S;
        if (runs <= kBlockThreads) return S;
    }
}

// Most runs the longest column (d terms) of the proposal sum is cut into, over all d served by tile count NT
__host__ __device__ constexpr int block_max_runs(int NT)
{
    int worst = 1;
    for (int d = 16 * (NT - 1) + 1; d <= 16 * NT; ++d) {
        const int S = block_run_length(d, 1, 4), r = (d + S - 1) / S;
        worst = r > worst ? r : worst;
    }
    return worst;
}

#ifndef __HIPCC_RTC__
// Cuts rows c = 0..d-1 with len(c) terms each into runs of at most S consecutive terms, S the smallest value that
// needs no more than kBlockThreads runs.  run r: (row[r], first[r], count[r]); rows appear in order.
struct BlockSegments {
    int S = 0;
    std::vector<int> row, first, count;
};
inline BlockSegments block_segments(int d, int extra, int min_run) // len(c) = c + extra (0: strictly below the diagonal, 1: with it)
{
    BlockSegments g;
    g.S = block_run_length(d, extra, min_run); // a run shorter than one batch of loads only adds threads, not speed
    for (int c = 0; c < d; ++c)
        for (int f = 0; f < c + extra; f += g.S) {
            g.row.push_back(c);
            g.first.push_back(f);
            g.count.push_back(std::min(g.S, c + extra - f));
        }
    return g;
}
#endif

template <int V>
struct DppCtrl {
    static constexpr int value = V;
};

// USER: the target is mcmc_user_target_part (run-time compiled module); its parameters take the place of Hq
// FAST: MCMC_FACTOR_RANK1 -- between full factorizations every refresh_every iterations, the factor follows covmat_empirical
// by the closed-form rank-1 update of fast_update below (not a parity mode; FULL and USER are off)
template <int NT, bool FULL, bool USER, bool FAST = false>
__global__ __launch_bounds__(kBlockThreads, NT <= 2 ? 4 : (NT <= 4 ? 3 : 2)) void rwm_block_kernel(const RwmArgs a, const double *__restrict__ prm)
{
    static_assert(!FAST || (!FULL && !USER), "the rank-1 mode has no diagnostics / injected streams / user targets");
    extern __shared__ double lds[];
    const int d = a.d;
    const BlockLayout L(d);
    const int T = L.T;
    int tid = threadIdx.x; // not const: made opaque once per iteration (see the top of the loop)
    double *const A = lds;        // packed upper factor R, (r,c) r<=c at c(c+1)/2 + r
    double *const th = A + T;     // theta_current
    double *const mn = th + d;    // theta_mean
    double *const Y4 = mn + d;    // proposal normals of four consecutive iterations (the four waves draw one each)
    double *const pr = Y4 + 4 * d; // theta_propose
    double *const v = pr + d;     // scratch: x - mean of the target, residual
    double *const sc = v + d + BlockLayout::kVPad; // v is followed by zeros; sc[12..15] = log(A) of four iterations; sc[4..7], sc[8..11]: per-wave partial sums
    double *const raw = sc + 16;  // two unscaled pivot rows (double buffer): 16 NT entries, the pivot, one pad
    constexpr int RS = 16 * NT + 2;
    double *const part = raw + 2 * RS + 16; // per-thread partial sums of the proposal; pivots / 1 / R_jj while R is factored
    // MCMC_FACTOR_RANK1 state (FX = 0 otherwise): rres = the last residual theta - theta_mean; qv = its whitened form
    // R^-T r with respect to the CURRENT factor; pv, gv, uv = the vectors of the update being applied; Rd = diag(R);
    // ptot = per-thread partial sums of the update
    const int FX = FAST ? L.fx() : 0;
    double *const rres = part + kBlockThreads;
    double *const qv = rres + FX, *const pv = qv + FX, *const gv = pv + FX, *const uv = gv + FX, *const Rd = uv + FX;
    double *const ptot = Rd + FX;
    double *const Hl = FAST ? ptot + kBlockThreads : part + kBlockThreads; // the first KL rows of Hq, [t][thread]
    constexpr int NB = NT * (NT + 1) / 2;
    const int64_t n = a.n_chains;
    // Workgroups go round-robin to the 8 XCDs, each with its own L2, and a chain writes 8 bytes of a 64-byte trace
    // line.  XCD x therefore takes a contiguous range of chains, so that the eight chains of a line are workgroups
    // dispatched together on ONE XCD and their stores merge in its L2 (measured with one chain per consecutive
    // workgroup: 257 GB written for a 52 GB trace).
    const int64_t c = [&]() -> int64_t {
        const int64_t x = blockIdx.x & 7, j = blockIdx.x >> 3, q = n >> 3, r = n & 7;
        return x * q + (x < r ? x : r) + j;
    }();
    const uint64_t gchain = (uint64_t)(a.chain_offset + c);
    const PhiloxKey key(a.seed);
    const bool inject = FULL && a.d_Z != nullptr;
    const int ass = a.adapt_size_start, ashs = a.adapt_shape_start;
    const int window = a.acceptance_window;
    const double wgt = a.acceptance_rate_weight;
    const bool need_moments = ashs != 0 || (FULL && a.d_sigma_trace) || a.d_final_mean || a.d_final_cov || a.d_state_in || a.d_state_out;
    const int i0 = a.iter_offset; // iterations done before this launch (checkpoint / resume)
    const double *__restrict__ tmean = prm + L.tmean();
    const double *__restrict__ Hd = prm + L.hdiag();
    const double *__restrict__ Hq = prm + L.hq();
    // this thread's runs of the target's and the proposal's triangular sums
    // Hq rows in LDS for NT = 3 .. 7 (below: the row kernel serves production and the runs are two loads long; NT = 8:
    // the chain's own state leaves no room).  At NT = 7 this only pays since the lane masks are no longer hoisted
    // (see the top of the iteration loop): before, the extra loop spilled and d = 100 ran 4.80 ms instead of 4.21
    // per 100 iterations of 4 096 chains; now 3.33 against 3.54 (size phase without size adaptation, measured).
    const int TP = __builtin_amdgcn_readfirstlane((int)prm[L.meta()]), KL = (NT >= 3 && NT <= 7) ? __builtin_amdgcn_readfirstlane((int)prm[L.meta() + 1]) : 0;
    int t_row = (int)prm[L.ttab() + 2 * tid];
    const int t_first = (int)prm[L.ttab() + 2 * tid + 1];
    int p_col = (int)prm[L.ptab() + 3 * tid];
    const int p_first = (int)prm[L.ptab() + 3 * tid + 1], p_len = (int)prm[L.ptab() + 3 * tid + 2];
    const int r_first = tid < d ? (int)prm[L.pred() + 2 * tid] : 0, r_count = tid < d ? (int)prm[L.pred() + 2 * tid + 1] : 0;
    const int ti = tid >> 4, tk = tid & 15; // tile row / column: the 16 owners of a row sit in one wave
    // this thread's entry of the target mean and of diag(H), in registers: read from L2 where they are used, each was a
    // round trip on the critical path of every iteration (the parameter block extends past d + 255: no range test)
    const double tm_k = USER ? 0.0 : tmean[tid], hd_k = USER ? 0.0 : Hd[tid];
    // FAST: the runs of a column are consecutive threads in ascending k; this thread's column ends at thread run_end - 1
    const int run_end = (FAST && p_col >= 0) ? (int)prm[L.pred() + 2 * p_col] + (int)prm[L.pred() + 2 * p_col + 1] : 0;

    // covmat_empirical tile (registers) and the working copy the factorization consumes
    // At doubles as W: between factorizations it holds this thread's run of the proposal sum -- SP consecutive entries
    // of one column of R -- so that an iteration reads only Y from LDS (R changes only when it is refactored)
    constexpr int WS = block_run_length(16 * NT, 1, 4); // longest proposal run any d of this NT can have
    constexpr int WN = NB > WS ? NB : WS;
    double Ct[NB], At[WN];
    double(&W)[WN] = At;
#pragma unroll
    for (int q = 0; q < NB; ++q) Ct[q] = 0.0;
    auto load_runs = [&]() { // after R in LDS changed and a barrier
#pragma unroll
        for (int u = 0; u < WN; ++u) W[u] = (p_col >= 0 && u < p_len) ? A[tri(0, p_col) + p_first + u] : 0.0;
    };
    for (int e = tid; e < T; e += kBlockThreads) A[e] = prm[L.r0() + e];
    if (tid < d) {
        const double x0 = a.d_init ? a.d_init[(int64_t)tid * n + c] : prm[L.init() + tid];
        th[tid] = x0;
        mn[tid] = x0;
        pr[tid] = 0.0; // the proposal runs multiply up to WS entries from Y + first on, zero coefficients past their end:
        v[tid] = 0.0;  // what follows a Y in LDS must be finite from the first iteration on (0 * NaN)
#pragma unroll
        for (int q = 0; q < 4; ++q) Y4[q * d + tid] = 0.0;
    }
    if (tid < BlockLayout::kVPad) v[d + tid] = 0.0;
    if constexpr (FAST) {
        for (int e = tid; e < 6 * FX; e += kBlockThreads) rres[e] = 0.0;
        ptot[tid] = 0.0;
    }
    for (int t = 0; t < KL; ++t) Hl[t * kBlockThreads + tid] = Hq[t * kBlockThreads + tid];
    __syncthreads();
    load_runs();
    if (FAST && tid < d) Rd[tid] = A[tri(tid, tid)];

    // sum over the wavefront, every lane gets it: four DPP steps inside each row of 16 lanes (no LDS round trip, unlike
    // a shuffle), then the four row totals through scalar registers
    auto wave_sum = [](double x) -> double {
        auto dpp_add = [](double y, auto ctrl) -> double {
            constexpr int C = decltype(ctrl)::value;
            const uint64_t b = (uint64_t)__double_as_longlong(y);
            // bound_ctrl: every lane has a source under these four controls, so no "old" value needs materialising
            const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, C, 0xf, 0xf, true);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), C, 0xf, 0xf, true);
            return y + __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
        };
        x = dpp_add(x, DppCtrl<0xB1>{});  // quad_perm [1,0,3,2]: lane ^ 1
        x = dpp_add(x, DppCtrl<0x4E>{});  // quad_perm [2,3,0,1]: lane ^ 2
        x = dpp_add(x, DppCtrl<0x141>{}); // row_half_mirror: the other quad of the same half row
        x = dpp_add(x, DppCtrl<0x140>{}); // row_mirror: the other half of the row
        auto row_total = [&](int lane) -> double {
            const uint64_t b = (uint64_t)__double_as_longlong(x);
            const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)b, lane), hi = __builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
            return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
        };
        return (row_total(0) + row_total(16)) + (row_total(32) + row_total(48));
    };
    auto wave_prod = [](double x) -> double { // the same tree with products
        auto dpp_mul = [](double y, auto ctrl) -> double {
            constexpr int C = decltype(ctrl)::value;
            const uint64_t b = (uint64_t)__double_as_longlong(y);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, C, 0xf, 0xf, true);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), C, 0xf, 0xf, true);
            return y * __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
        };
        x = dpp_mul(x, DppCtrl<0xB1>{});
        x = dpp_mul(x, DppCtrl<0x4E>{});
        x = dpp_mul(x, DppCtrl<0x141>{});
        x = dpp_mul(x, DppCtrl<0x140>{});
        auto row_total = [&](int lane) -> double {
            const uint64_t b = (uint64_t)__double_as_longlong(x);
            const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)b, lane), hi = __builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
            return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
        };
        return (row_total(0) * row_total(16)) * (row_total(32) * row_total(48));
    };
    // inclusive prefix sum over the lanes of a wavefront: four row_shr steps inside each row of 16, then lane 15 of a row
    // into the next row (rows 1, 3), then lane 31 into rows 2, 3
    [[maybe_unused]] auto wave_scan = [](double x) -> double {
        auto step = [](double y, auto ctrl, auto rmask) -> double {
            constexpr int C = decltype(ctrl)::value, RM = decltype(rmask)::value;
            const uint64_t b = (uint64_t)__double_as_longlong(y);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, C, RM, 0xf, false);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), C, RM, 0xf, false);
            return y + __longlong_as_double((long long)(((uint64_t)hi << 32) | lo)); // lanes without a source add +0.0
        };
        x = step(x, DppCtrl<0x111>{}, DppCtrl<0xf>{}); // row_shr:1
        x = step(x, DppCtrl<0x112>{}, DppCtrl<0xf>{}); // row_shr:2
        x = step(x, DppCtrl<0x114>{}, DppCtrl<0xf>{}); // row_shr:4
        x = step(x, DppCtrl<0x118>{}, DppCtrl<0xf>{}); // row_shr:8
        x = step(x, DppCtrl<0x142>{}, DppCtrl<0xa>{}); // row_bcast:15 into rows 1 and 3
        x = step(x, DppCtrl<0x143>{}, DppCtrl<0xc>{}); // row_bcast:31 into rows 2 and 3
        return x;
    };
    // -0.5 (x - m)^T H (x - m); result replicated in every thread.  An iteration is a chain of latencies, not of
    // flops: a wave alone on its SIMD issues one instruction per ~11 cycles (measured with per-phase clocks at
    // d = 100: 12 instructions and an exposed LDS read per term made this sum 6 000 cycles), so what counts is the
    // number of instructions per term and per thread.  Each thread owns one run of consecutive terms of one row
    // (block_segments): its coefficients come coalesced from L2, eight loads in flight, the v entries from LDS at
    // constant offsets, one FMA per term; one wave reduction plus four partial sums in LDS finish the sum.
    // Precondition: a barrier since v[k] = x[k] - mean[k] (USER: x[k]) was stored by thread k for every k < d.
    auto target = [&](const double *x, auto &&mid) -> double {
        if constexpr (USER) { // every thread one part of the user's sum; x (theta or the proposal) is complete in LDS
            mid();
#ifdef MCMC_USER_BLOCK
            const double ws = wave_sum(::mcmc_user_target_part(x, d, Hq, tid, kBlockThreads));
            if ((tid & 63) == 0) sc[4 + (tid >> 6)] = ws;
            __syncthreads();
            return (sc[4] + sc[5]) + (sc[6] + sc[7]);
#else
            return 0.0;
#endif
        }
        // Every thread runs the sum, with or without a run (no exec-mask branch: a thread without one has zero coefficients
        // and its result is dropped by the select at the end), and no load is guarded: Hq is followed by kHqPad zero rows.
        // `mid` is work that does not depend on the sum (the Hastings term's exp / log chain): it is placed behind the
        // first coefficient requests, so that it runs while they are on their way.
        double acc = 0.0;
        {
            const double *const vp = v + t_first;
            const double *const hp = Hq + tid;
            constexpr int CH = NT <= 4 ? 4 : 8; // loads in flight; small d: short runs, and the registers decide the occupancy
            static_assert(CH <= BlockLayout::kHqPad, "the zero rows behind Hq cover one batch of loads");
            // Coefficients KL .. KL + CH - 1 are requested from L2 first and used last; the rows before them sit in LDS
            // (the L2 round trips of this sum, TP / CH of them, were 19 % of an iteration at d = 100; measured)
            int t_l2 = 0;
            if constexpr (NT >= 3 && NT <= 7) {
                double hl[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) hl[u] = hp[(KL + u) * kBlockThreads];
                mid();
                const double *const hs = Hl + tid;
                for (int t = 0; t < KL; t += 4) { // KL is a multiple of 4
                    double h[4], vv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        h[u] = hs[(t + u) * kBlockThreads];
                        vv[u] = vp[t + u];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = fma(h[u], vv[u], acc);
                }
                double vv[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) vv[u] = vp[KL + u];
#pragma unroll
                for (int u = 0; u < CH; ++u) acc = fma(hl[u], vv[u], acc);
                t_l2 = KL + CH;
            } else {
                mid();
            }
            for (int t = t_l2; t < TP; t += CH) {
                double h[CH], vv[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) h[u] = hp[(t + u) * kBlockThreads];
#pragma unroll
                for (int u = 0; u < CH; ++u) vv[u] = vp[t + u]; // past the run: zero coefficient, finite v or the zero pad
#pragma unroll
                for (int u = 0; u < CH; ++u) acc = fma(h[u], vv[u], acc);
            }
            const double vr = v[t_row < 0 ? 0 : t_row];
            acc = (t_row >= 0) ? acc * vr : 0.0;
        }
        {   // the diagonal term: unguarded reads (the parameter block and the LDS behind v extend past d + 255) and a select
            const double vd = v[tid];
            const double accd = fma(hd_k * vd, vd, acc);
            acc = (tid < d) ? accd : acc;
        }
        const double ws = wave_sum(acc);
        if ((tid & 63) == 0) sc[4 + (tid >> 6)] = ws;
        __syncthreads();
        return -0.5 * ((sc[4] + sc[5]) + (sc[6] + sc[7]));
    };

    double rootTerm = prm[L.root0()];
    double prodR = 1.0; // product of diag(R)
    double min_var = prm[L.cov()]; // smallest diagonal entry of covmat: decides the test at :126 (rounding is monotone)
    for (int k = 0; k < d; ++k) {
        prodR *= prm[L.r0() + tri(k, k)];
        min_var = fmin(min_var, prm[L.cov() + tri(k, k)]);
    }
    const double prodR0 = prodR;
    bool base_emp = false; // R holds chol(covmat_empirical)
    double s_eff = 1.0;    // covmat_proposal = s_eff^2 R^T R
    // FAST: R (the runs W, the diagonal Rd) factors the covmat_empirical the LAST shape-phase iteration proposed from, and
    // qv holds q = R^-T r of the residual r that iteration's update_sigma used: what the rank-1 update needs
    [[maybe_unused]] bool fast_valid = false;
    [[maybe_unused]] int refresh_ctr = FAST ? (a.iter_offset % a.refresh_every) : 0; // (i - 1) mod refresh_every at the top of iteration i
    if (!USER && tid < d) v[tid] = th[tid] - tm_k;
    __syncthreads();
    double t_cur = target(th, [] {});
    double rate = 0.0, scaling_sd = 1.0, scaling_multiplier = 1.0;
    AcceptWindow win; // replicated in every thread, like the other scalars of the chain; thread 0 stores the ring words
    int win_pos = accept_window_pos(i0, window);
    unsigned long long *const ring = a.d_win_ring ? a.d_win_ring + c : nullptr;
    int32_t status = MCMC_CHAIN_OK, n_shape = 0;
    bool dead = false;
    const RwmStateLayout SL = rwm_state_layout(d);
    if (a.d_state_in) { // resume: everything the loop carries from one iteration to the next
        const double *const sp = a.d_state_in + c;
        __syncthreads();
        if (tid < d) {
            th[tid] = sp[(int64_t)(SL.THETA + tid) * n];
            mn[tid] = sp[(int64_t)(SL.MEAN + tid) * n];
        }
        for (int e = tid; e < T; e += kBlockThreads) A[e] = sp[(int64_t)(SL.R + e) * n];
#pragma unroll
        for (int bb = 0; bb < NT; ++bb) {
#pragma unroll
            for (int aa = 0; aa <= bb; ++aa) {
                const int r = 16 * aa + ti, cl = 16 * bb + tk;
                if (r <= cl && cl < d) Ct[blk(aa, bb)] = sp[(int64_t)(SL.C + tri(r, cl)) * n];
            }
        }
        const double *const ss = sp + (int64_t)SL.S * n;
        t_cur = ss[ST_TCUR * n];
        rate = ss[ST_RATE * n];
        scaling_sd = ss[ST_SD * n];
        scaling_multiplier = ss[ST_MULT * n];
        s_eff = ss[ST_SEFF * n];
        prodR = ss[ST_PRODR * n];
        rootTerm = ss[ST_ROOT * n];
        win.bits = ((uint64_t)(uint32_t)ss[ST_WHI * n] << 32) | (uint64_t)(uint32_t)ss[ST_WLO * n];
        win.len = (int)ss[ST_WLEN * n];
        win.count = (int)ss[ST_WCOUNT * n];
        for (int wd = tid; wd < a.win_words; wd += kBlockThreads) // the outcome ring (acceptance_window > 64) follows the scalars
            ring[(int64_t)wd * n] = ((uint64_t)(uint32_t)sp[(int64_t)(SL.TOTAL + 2 * wd + 1) * n] << 32) |
                                    (uint64_t)(uint32_t)sp[(int64_t)(SL.TOTAL + 2 * wd) * n];
        base_emp = ss[ST_BASE * n] != 0.0;
        if constexpr (FAST) {
            fast_valid = ss[ST_FAST * n] != 0.0;
            if (tid < d) {
                rres[tid] = sp[(int64_t)(SL.Q + tid) * n];
                qv[tid] = sp[(int64_t)(SL.Q + d + tid) * n];
            }
        }
        status = (int32_t)ss[ST_STATUS * n];
        n_shape = (int32_t)ss[ST_NSHAPE * n];
        dead = status != MCMC_CHAIN_OK;
        if (ss[ST_TAG * n] != (double)MCMC_KERNEL_BLOCK || ss[ST_ITER * n] != (double)i0) { // not a state this launch can continue
            dead = true;
            status = ((i0 + 1) << 4) | MCMC_CHAIN_BAD_STATE;
            if (tid < d) th[tid] = __builtin_nan("");
        }
        __syncthreads();
        load_runs();
        if (FAST && tid < d) Rd[tid] = A[tri(tid, tid)];
    }

    // pow(adapt_size_cooling, i - adapt_size_start), fetched one iteration ahead: the adaptation opens the
    // iteration and would otherwise wait for an L2 round trip
    double cool_next = ass != 0 ? a.d_cool_pow[min(1, a.cool_last)] : 0.0;
    int thin_ctr = i0 % a.thin; // i mod thin and the number of rows kept so far, as running counters
    int64_t t_kept = 0;
    for (int i = i0 + 1; i <= i0 + a.n_iterations; ++i) {
        const int li = i - i0; // position inside this launch: streams, diagnostics and the cooling table are per launch
        const double di = (double)i;
        const double cool = cool_next;
        // The lane masks of this loop (tid < d, tid == 0, first lane of a wave, "has a run") are loop invariants: left
        // alone the compiler computes ~30 of them once, keeps them in scalar registers it does not have, and fetches
        // each back with two v_readlane per use.  Opaque values make it form a mask where it is used (one v_cmp).
        // Only where registers are short (NT >= 7: 44 bytes of scratch gone, d = 100 +4 %); at NT = 5, 6 the recomputed
        // masks and addresses cost 5 % (measured).
        if constexpr (NT >= 7) asm volatile("" : "+v"(tid), "+v"(t_row), "+v"(p_col));
        if (ass != 0 && li < a.n_iterations) cool_next = a.d_cool_pow[min(li + 1, a.cool_last)];
        // ---- adaptation of covmat_proposal (adaptMCMC_source.cpp:117-139), replicated ---------------
        bool rescale = false, refactor = false;
        if (ass != 0 && i >= ass && (ashs == 0 || mul_rn(rate, di) < (double)ashs)) {
            // polynomial coefficients by one scalar table load at the use (fastmath.h: fm_tab), not 2 s_mov_b32 each
            scaling_multiplier = fm_exp_t(cool * (rate - 0.234), fm_tab<16>(FT_EXP, i));
            scaling_sd = mul_rn(scaling_sd, scaling_multiplier);
            scaling_sd = (a.max_scaling_sd < scaling_sd) ? a.max_scaling_sd : scaling_sd;
            const double s2 = mul_rn(scaling_sd, scaling_sd);
            const bool any_small = mul_rn(s2, min_var) < 2E-16;
            rescale = !any_small && !dead; // covmat_proposal = scaling_sd^2 covmat (:129)
        } else if (ashs != 0 && mul_rn(rate, di) >= (double)ashs) {
            scaling_sd = a.shape_scaling_sd;
            refactor = !dead; // covmat_proposal = scaling_sd^2 covmat_empirical (:138)
            ++n_shape;
        }
        // FAST: a shape-phase iteration whose factor is one rank-1 step behind covmat_empirical catches up by the update
        // instead of a factorization, unless this iteration is a refresh point (absolute iteration numbers)
        [[maybe_unused]] const bool shape_it = refactor;
        [[maybe_unused]] bool fast_step = false;
        if constexpr (FAST) {
            const bool refresh_due = ++refresh_ctr == a.refresh_every; // i % refresh_every == 0
            if (refresh_due) refresh_ctr = 0;
            fast_step = shape_it && fast_valid && !refresh_due;
            refactor = shape_it && !fast_step;
        }
        if (FULL && tid == 0) {
            if (a.d_scaling_sd) a.d_scaling_sd[(int64_t)(li - 1) * n + c] = scaling_sd;
            if (a.d_scaling_multiplier) a.d_scaling_multiplier[(int64_t)(li - 1) * n + c] = scaling_multiplier;
        }

        // ---- R = chol(covmat_proposal) when it was assigned (:153) --------------------------------
        if (rescale) { // uniform over the workgroup
            if (base_emp) { // back from the shape phase: the factor of covmat again
                __syncthreads();
                for (int e = tid; e < T; e += kBlockThreads) A[e] = prm[L.r0() + e];
                if (FAST && tid < d) Rd[tid] = prm[L.r0() + tri(tid, tid)];
                __syncthreads();
                load_runs();
                prodR = prodR0;
                base_emp = false;
            }
            s_eff = scaling_sd;
        }
        if constexpr (FAST) {
            if (fast_step) { // uniform over the workgroup
                // covmat_empirical_{i-1} = a C_{i-2} + b r r^T (update_sigma, :43, with a = (i-2)/(i-1), b = a/(i-1)) and
                // C_{i-2} = R^T R, so C_{i-1} = a R^T (I + p p^T) R with p = sqrt(b/a) q, q = R^-T r (qv).  The upper
                // factor of I + p p^T is known in closed form -- diagonal sqrt(sg_{k+1} / sg_k), entry (k, j > k)
                // p_k p_j / sqrt(sg_k sg_{k+1}), sg_k = 1 + sum_{i<k} p_i^2 -- hence
                //     R'_kj = g_k R_kj + u_k sum_{k < i <= j} p_i R_ij,   g_k = sqrt(a sg_{k+1} / sg_k),  u_k = sqrt(a) p_k / sqrt(sg_k sg_{k+1})
                // and  R'^-T r = q_k / (sqrt(a) sqrt(sg_k sg_{k+1})): one prefix sum over d numbers, d square roots in
                // parallel, and per column of R a suffix sum -- no recurrence over the pivots, O(d^2) flops.
                const double dj = di - 1.0; // the update belongs to iteration i - 1
                const double aj = (dj - 1.0) / dj;
                double sqa, rsqa, sqj, rsqj;
                fm_sqrt_rsq(aj, sqa, rsqa);
                fm_sqrt_rsq(dj, sqj, rsqj); // sqrt(b / a) = 1 / sqrt(i - 1)
                const double qk = tid < d ? qv[tid] : 0.0;
                const double pk = qk * rsqj, p2 = pk * pk;
                double inc = wave_scan(p2);
                if ((tid & 63) == 63) sc[tid >> 6] = inc; // the wave totals (waves 2, 3 hold zeros: d <= 128)
                __syncthreads();
                const double tot0 = sc[0], tot1 = sc[1];
                if (tid >= 64) inc += tot0;
                const double sg = 1.0 + (inc - p2), sg1 = 1.0 + inc;
                double rt_unused, rs;
                fm_sqrt_rsq(sg * sg1, rt_unused, rs);
                if (tid < d) {
                    const double g = sqa * sg1 * rs;
                    gv[tid] = g;
                    uv[tid] = sqa * pk * rs;
                    pv[tid] = pk;
                    qv[tid] = qk * rs * rsqa; // R'^-T r
                    Rd[tid] *= g;             // the diagonal: R'_kk = g_k R_kk
                }
                {   // prod diag(R') = prod diag(R) sqrt(a)^d sqrt(sg_d)
                    double sp = 1.0, b = sqa;
                    for (int e = d; e > 0; e >>= 1) {
                        if (e & 1) sp *= b;
                        b *= b;
                    }
                    double st, su_unused;
                    fm_sqrt_rsq(1.0 + (tot0 + tot1), st, su_unused);
                    prodR = prodR * sp * st;
                }
                __syncthreads();
                // this thread's run of column c: first its share of the suffix sums, then the carry of the runs above it
                // LDS operands in batches, requested together (as in the proposal run below); threads without a run take part
                // with W = 0 and their results are never read
                constexpr int FB = 8;
                double tot = 0.0;
                {
                    const double *const pp = pv + p_first; // past the run: W is zero, pv finite (zero past d)
#pragma unroll
                    for (int b0 = 0; b0 < WS; b0 += FB) {
                        double pb[FB];
#pragma unroll
                        for (int u = 0; u < FB; ++u) pb[u] = (b0 + u < WS) ? pp[b0 + u] : 0.0;
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < FB; ++u)
                            if (b0 + u < WS) tot = fma(pb[u], W[b0 + u], tot);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                ptot[tid] = tot;
                __syncthreads();
                {
                    // sum_{k < i <= c} p_i R_ic for the entry being formed: first the totals of the runs above this one in
                    // the column (a fixed number of unguarded reads and selects; ptot is followed by the Hq rows or nothing
                    // past thread 255, hence the clamp)
                    constexpr int RMAX = block_max_runs(NT);
                    double above[RMAX];
#pragma unroll
                    for (int q = 0; q < RMAX - 1; ++q) above[q] = ptot[min(tid + 1 + q, kBlockThreads - 1)];
                    double sfx = 0.0;
#pragma unroll
                    for (int q = 0; q < RMAX - 1; ++q) sfx = (tid + 1 + q < run_end) ? sfx + above[q] : sfx;
                    const double *const pp = pv + p_first, *const gp = gv + p_first, *const up = uv + p_first;
                    constexpr int NBAT = (WS + FB - 1) / FB;
#pragma unroll
                    for (int bi = NBAT - 1; bi >= 0; --bi) {
                        const int b0 = bi * FB;
                        double pb[FB], gb[FB], ub[FB];
#pragma unroll
                        for (int u = 0; u < FB; ++u) {
                            const bool in = b0 + u < WS;
                            pb[u] = in ? pp[b0 + u] : 0.0;
                            gb[u] = in ? gp[b0 + u] : 0.0;
                            ub[u] = in ? up[b0 + u] : 0.0;
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = FB - 1; u >= 0; --u) {
                            if (b0 + u < WS) {
                                const double w = W[b0 + u];
                                const double nw = fma(gb[u], w, ub[u] * sfx);
                                sfx = fma(pb[u], w, sfx);
                                W[b0 + u] = (b0 + u < p_len) ? nw : 0.0;
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                s_eff = scaling_sd;
            }
        }
        if (refactor) { // uniform over the workgroup
#pragma unroll
            for (int q = 0; q < NB; ++q) At[q] = Ct[q];
            // FAST: the residual rides along as an extra column of the matrix being factored, which leaves R^-T r in it
            [[maybe_unused]] double ek = (FAST && tid < d) ? rres[tid] : 0.0;
            bool ok = true;
            // this thread's two windows into the row buffers, as bases of their own (opaque: merged with a neighbouring LDS
            // pointer, as the compiler does, the second buffer's entries fall outside the 8-bit offset fields of the LDS
            // instructions and every odd pivot pays four address additions)
            typedef __attribute__((address_space(3))) double LdsDouble; // 32-bit LDS addresses: one register each
            LdsDouble *rtk = (LdsDouble *)(raw + tk), *rti = (LdsDouble *)(raw + ti);
            asm volatile("" : "+v"(rtk), "+v"(rti));
#pragma unroll
            for (int a0 = 0; a0 < NT; ++a0) {
                const int jt_end = __builtin_amdgcn_readfirstlane(min(16, d - 16 * a0)); // one scalar loop bound, not a range test per pivot
                // two pivots per turn of the loop, so that WHICH of the two row buffers a pivot uses is known at compile time
                // and lands in the offset fields of its LDS instructions (selected per pivot it was three scalar and four
                // vector instructions of address arithmetic in every wave)
                for (int jt0 = 0; jt0 < jt_end; jt0 += 2) {
#pragma unroll
                  for (int par = 0; par < 2; ++par) {
                    const int jt = jt0 + par;
                    if (par == 1 && jt >= jt_end) break; // uniform
                    const int j = 16 * a0 + jt;
                    double *const row = raw + par * RS; // j & 1 == par: 16 a0 and jt0 are even
                    if (ti == jt) { // publish row j: zero up to and including the pivot, so that readers need no range test
#pragma unroll
                        for (int bb = a0; bb < NT; ++bb) {
                            const double val = At[blk(a0, bb)]; // entries beyond d are exactly zero
                            if (bb == a0) {
                                if (tk == jt) {
                                    row[16 * NT] = val;
                                    part[j] = val; // the pivot, for the square roots after the loop (by its owner: a store of
                                                   // thread 0's own cost every wave a masked block per pivot)
                                }
                                rtk[par * RS + 16 * bb] = (tk > jt) ? val : 0.0;
                            } else {
                                rtk[par * RS + 16 * bb] = val;
                            }
                        }
                    }
                    if constexpr (FAST) {
                        if (tid == j) row[16 * NT + 1] = ek; // entry j of the extra column, finished by the pivots before j
                    }
                    __syncthreads();
                    // everything this pivot reads is requested here, together (left alone the compiler waits for the pivot,
                    // forms its reciprocal and only then requests the row, a pair at a time: four exposed LDS round trips)
                    const double ajj = row[16 * NT]; // (whether it is positive is looked at once, after the loop, in part[])
                    double ri[NT], rk[NT];
#pragma unroll
                    for (int x = a0; x < NT; ++x) {
                        ri[x] = rti[par * RS + 16 * x];
                        rk[x] = rtk[par * RS + 16 * x];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const double inv = fm_rcp(ajj);
                    if constexpr (FAST) { // e_k -= (a_jk / a_jj) e_j for k > j (the published row starts at tile column a0
                                          // and is zero up to the pivot; what lies before 16 a0 is an older row's)
                        if (tid >= 16 * a0 && tid < 16 * NT) ek = fma(-(row[tid] * inv), row[16 * NT + 1], ek);
                    }
#pragma unroll
                    for (int x = a0; x < NT; ++x) ri[x] *= inv;
                    // Nothing else happens to row j here: it stays in the registers of its owners -- the published row is zero
                    // up to and including the pivot, so for the finished rows ri is exactly 0 and the updates below leave
                    // them as they are -- and goes to its place in A after the loop, scaled, with all the other rows
                    // (copied out here, one entry per thread of the first waves, it was a masked block with an LDS round
                    // trip per pivot: ~10 of ~90 instructions).  The square roots wait until all pivots are done too.
                    // (With the owners finishing their row here -- square root, NT scaled stores -- one wave ran ~60
                    // instructions more than the others at every pivot, and the barrier waited for it: 32 % of a
                    // shape-phase run at d = 100, measured.)
#pragma unroll
                    for (int bb = a0; bb < NT; ++bb) {
#pragma unroll
                        for (int aa = a0; aa <= bb; ++aa) At[blk(aa, bb)] = fma(-ri[aa], rk[bb], At[blk(aa, bb)]);
                    }
                  }
                }
            }
            __syncthreads();
            {   // R_jj = sqrt(pivot_j), all rows at once; part[j] becomes 1 / R_jj
                // arma::chol fails on the first pivot that is not positive (:153): every pivot is in part[] (one that is
                // zero, negative or NaN leaves garbage behind it, but stays there itself), tested here once instead of
                // two instructions per pivot in every wave; the verdict travels with the diagonal product
                double rjj = 1.0;
                const double pvt = tid < d ? part[tid] : 1.0;
                const bool wave_bad = __builtin_amdgcn_ballot_w64(!(pvt > 0.0)) != 0;
                if ((tid & 63) == 0) sc[8 + (tid >> 6)] = wave_bad ? 1.0 : 0.0; // (the |Y|^2 slots: free until the proposal)
                if (tid < d) {
                    double rinv;
                    fm_sqrt_rsq(pvt, rjj, rinv);
                    A[tri(tid, tid)] = rjj;
                    part[tid] = rinv;
                    if constexpr (FAST) {
                        Rd[tid] = rjj;
                        qv[tid] = ek * rinv; // R^-T r
                    }
                }
                const double wp = wave_prod(rjj);
                if ((tid & 63) == 0) sc[4 + (tid >> 6)] = wp;
            }
            __syncthreads();
            prodR = (sc[4] * sc[5]) * (sc[6] * sc[7]);
            ok = (sc[8] + sc[9]) + (sc[10] + sc[11]) == 0.0;
            // the rows of the factor, R_rc = (unscaled entry) / R_rr, from the registers of the tile layout to their places in A
            {
                double rinv[NT];
#pragma unroll
                for (int aa = 0; aa < NT; ++aa) rinv[aa] = part[16 * aa + ti]; // 1 / R_rr of this thread's rows (16 NT <= 256 entries)
#pragma unroll
                for (int bb = 0; bb < NT; ++bb) {
#pragma unroll
                    for (int aa = 0; aa <= bb; ++aa) {
                        const int r = 16 * aa + ti, cl = 16 * bb + tk;
                        if (r < cl && cl < d) A[tri(r, cl)] = At[blk(aa, bb)] * rinv[aa];
                    }
                }
            }
            __syncthreads();
            load_runs(); // every thread takes the run of its proposal sum and keeps it in registers (W)
            __syncthreads();
            base_emp = true;
            s_eff = scaling_sd;
            if (!ok) { // arma::chol throws here; the chain stops and reports where
                dead = true;
                status = (i << 4) | MCMC_CHAIN_CHOL_FAILED;
                if (tid < d) th[tid] = __builtin_nan("");
                __syncthreads();
            }
        }
        // rootTerm is used by the last wave alone (it forms the Hastings term, below): only that wave keeps it up to date
        if ((rescale || refactor || (FAST && fast_step)) && tid >= kBlockThreads - 64) { // 1 / sqrt((2 pi)^d det(covmat_proposal)) (:56)
            double sp = 1.0, b = s_eff;
            for (int e = d; e > 0; e >>= 1) { // s_eff^d by repeated squaring
                if (e & 1) sp *= b;
                b *= b;
            }
            const double dt = sp * prodR;
            double det = dt * dt;
            if (!(det > 1e-280 && det < 1e280)) {
                // Near under / overflow the way the product is formed decides WHEN it reaches 0 or Inf, and from that
                // iteration on the reference rejects every proposal (rootTerm = Inf or 0 -> p = +-Inf -> la = NaN, :56,
                // :169-175; at d = 100 with covmat = I this happens once scaling_sd < 0.0241).  arma::det is the running
                // product of the LU diagonal, here diag(U)_k = s^2 R_kk^2: the same product in the same order.
                const double s2 = mul_rn(s_eff, s_eff);
                det = 1.0;
                for (int k = 0; k < d && det != 0.0; ++k) { // once 0 it stays 0 (every factor is finite)
                    const double rkk = FAST ? Rd[k] : A[tri(k, k)];
                    det = mul_rn(det, mul_rn(s2, mul_rn(rkk, rkk)));
                }
            }
            double sq_unused;
            fm_sqrt_rsq(a.twopi_pow_d * det, sq_unused, rootTerm); // det = 0 or Inf gives NaN here, Inf / 0 in the reference: la is NaN either way
        }

        // ---- draws: d normals, then one uniform (:149-152,174) ----------------------------------------
        const int slot = inject ? 0 : ((li - 1) & 3);
        double *const Y = Y4 + slot * d;
        if (inject) {
            if (tid < d) Y[tid] = a.d_Z[((int64_t)(li - 1) * d + tid) * n + c];
            if (tid == 0) sc[12] = fm_log<true>(a.d_U[(int64_t)(li - 1) * n + c]);
            __syncthreads();
        } else if (slot == 0) {
            // Every fourth iteration the four waves draw for four iterations, wave w for iteration i + w: a Philox block
            // and a Box-Muller pair are ~200 dependent instructions whichever lanes take part (13 % of an iteration at
            // d = 100 when one wave drew every iteration and three waited; measured).  Lane l < nb draws normals 2l,
            // 2l + 1, lane nb the uniform -- the same code: log(u0) is the first step of Box-Muller.
            const int wv = tid >> 6, ln = tid & 63, nb = (d + 1) / 2;
            PhiloxKey kk = key; // opaque copy: the round keys are formed by scalar adds here, not hoisted and spilled
            asm volatile("" : "+s"(kk.k0), "+s"(kk.k1));
            const ScalarCoef<16> tl = fm_tab<16>(FT_LOG, i), ts = fm_tab<16>(FT_SINCOS, i);
            const uint32_t it = (uint32_t)(i + wv); // past the last iteration of the launch: drawn, never used
            double *const Yw = Y4 + wv * d;
            if (ln <= nb) {
                const bool uni = ln == nb;
                double z0, z1, lg;
                philox_normal2_log_t(kk, gchain, it, uni ? TAG_RWM_UNIFORM : TAG_RWM_NORMAL, uni ? 0u : (uint32_t)ln, z0, z1, lg, tl, ts);
                if (uni) {
                    sc[12 + wv] = lg;
                } else {
                    Yw[2 * ln] = z0;
                    if (2 * ln + 1 < d) Yw[2 * ln + 1] = z1;
                }
            }
            if (nb == 64 && ln == 0) { // d = 127, 128: the 64 lanes are all taken by normals
                double z0, z1, lg;
                philox_normal2_log_t(kk, gchain, it, TAG_RWM_UNIFORM, 0u, z0, z1, lg, tl, ts);
                sc[12 + wv] = lg;
            }
            __syncthreads();
        }

        // ---- proposal theta + Y * R (:153-154): one run of consecutive k of one column per thread, then the runs
        // of a column are added in order by its thread ------------------------------------------------------------
        {
            // every thread, with or without a run (no exec-mask branch): W is zero past the run and in threads without one
            // (load_runs); the Y read there is some finite entry of this workgroup's LDS
            // Eight Y entries are requested at a time: left alone the compiler reads two, waits, uses them, reads two more
            // -- WS / 2 exposed LDS round trips per run.  The scheduling fences keep the batches apart.
            double s = 0.0;
            {
                const double *const yp = Y + p_first;
                constexpr int PB = 8;
#pragma unroll
                for (int b0 = 0; b0 < WS; b0 += PB) {
                    double y[PB];
#pragma unroll
                    for (int u = 0; u < PB; ++u) y[u] = (b0 + u < WS) ? yp[b0 + u] : 0.0;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < PB; ++u)
                        if (b0 + u < WS) s = fma(y[u], W[b0 + u], s);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            part[tid] = s;
            // |Y|^2 for the Hastings term (unguarded read -- tid >= d reads what follows Y in LDS -- and a select)
            const double yt = Y[tid];
            const double yy = (tid < d) ? yt * yt : 0.0;
            const double ys = wave_sum(yy);
            if ((tid & 63) == 0) sc[8 + (tid >> 6)] = ys;
            __syncthreads();
            if (tid < d) { // thread k is the only one that touches theta[k], proposal[k] and writes v[k]
                // the runs of column tid, added in order: a fixed number of unguarded reads and selects (a loop with a
                // run-time count waits for every read in turn)
                constexpr int RMAX = block_max_runs(NT); // most runs a column can have at any d of this NT
                double pq[RMAX];
#pragma unroll
                for (int q = 0; q < RMAX; ++q) pq[q] = part[min(r_first + q, kBlockThreads - 1)];
                double t = pq[0];
#pragma unroll
                for (int q = 1; q < RMAX; ++q) t = (q < r_count) ? t + pq[q] : t;
                const double p = th[tid] + s_eff * t;
                pr[tid] = p;
                if (!USER) v[tid] = p - tm_k;
            }
        }
        const double logA = sc[12 + slot];
        const double q = (sc[8] + sc[9]) + (sc[10] + sc[11]);
        __syncthreads();

        // ---- target and Metropolis-Hastings log ratio (:157-171) --------------------------------------
        // The Hastings term log(rootTerm exp(-q/2)) (:48-64) depends on the draws only: formed INSIDE the target sum, behind
        // its first coefficient requests, so that its ~50 dependent instructions run while those are on their way
        // ONE wave forms it (the last: at d <= 128 it owns no coordinate) and publishes it through sc[3] under the sum's own
        // barrier: replicated in all four waves the chain was 60 of ~690 instructions per wave-iteration.
        const double t_prop = target(pr, [&] {
            if (tid >= kBlockThreads - 64) { // wave-uniform
                const double pe = rootTerm * fm_exp_t(-0.5 * q, fm_tab<16>(FT_EXP, li));
                double pp = fm_log_t(pe, fm_tab<16>(FT_LOG, li));
                pp = (pe == 0.0) ? -__builtin_inf() : pp; // the special cases of fm_log<true>: the NaN-reject path needs them
                pp = (pe < 0.0 || pe != pe) ? __builtin_nan("") : pp;
                pp = (pe == __builtin_inf()) ? pe : pp;
                if (tid == kBlockThreads - 64) sc[3] = pp;
            }
        });
        const double p = sc[3];
        double la;
        if (!isfinite(t_prop)) {
            la = -__builtin_inf();
        } else {
            la = t_prop - t_cur;
            la = add_rn(la, p);
            la = add_rn(la, -p);
        }
        const bool accept = logA < la;
        if (accept) {
            if (tid < d) th[tid] = pr[tid];
            t_cur = t_prop;
        }
        if (FULL && tid == 0) {
            if (a.d_accepted) a.d_accepted[(int64_t)(li - 1) * n + c] = accept ? 1 : 0;
            if (a.d_log_acceptance) a.d_log_acceptance[(int64_t)(li - 1) * n + c] = la;
            if (a.d_log_A) a.d_log_A[(int64_t)(li - 1) * n + c] = logA;
        }
        // no barrier: theta[k], theta_mean[k] and proposal[k] are read and written by thread k alone

        // ---- trace (:185) -------------------------------------------------------------------------------
        const bool keep = ++thin_ctr == a.thin; // i % thin == 0, without a division per iteration
        const int64_t t = t_kept;               // row inside this launch's trace
        if (keep) {
            thin_ctr = 0;
            ++t_kept;
        }
        if (a.d_trace && keep && tid < d) {
            if (a.trace_chain_major) // [chain][t][k]: the kept row of this chain is d contiguous doubles -- whole lines from ONE workgroup
                a.d_trace[((int64_t)c * a.n_kept + t) * d + tid] = th[tid];
            else
                a.d_trace[(t * d + tid) * n + c] = th[tid];
        }

        // ---- acceptance rate (:188-207), replicated ------------------------------------------------------
        const double acc = accept ? 1.0 : 0.0;
        if (i == 1) {
            rate = acc;
        } else if (wgt == 0) {
            if (window == 0) {
                rate = add_rn(rate, div_rn(add_rn(acc, -rate), di));
            } else {
                rate = accept_window_push(win, accept, window, win_pos, ring, n, tid == 0);
            }
        } else {
            rate = add_rn(mul_rn(rate, 1 - wgt), mul_rn(acc, wgt));
        }
        if (FULL && a.d_rate_trace && tid == 0) a.d_rate_trace[(int64_t)(li - 1) * n + c] = rate;

        // ---- running mean / covariance (:210-214, update_sigma :42-45) ----------------------------------
        if (need_moments) {
            const double inv_i = 1.0 / di;
            const double sfac = (di - 1.0) * inv_i; // update_sigma (:43) as C (i-1)/i + ((i-1)/i^2 r_a) r_b
            const double bfac = sfac * inv_i;
            if (tid < d) {
                const double mk = mn[tid];
                const double r = th[tid] - mk;
                v[tid] = r;
                mn[tid] = mk + r * inv_i;
                if constexpr (FAST) {
                    rres[tid] = r;
                    // q_i = R^-T r_i for the factor this iteration proposed from: r_i = (1 - 1/(i-1)) r_{i-1} + (theta_i -
                    // theta_{i-1}), and the accepted step is s_eff Y R, i.e. s_eff Y once whitened
                    if (shape_it) qv[tid] = fma(1.0 - 1.0 / (di - 1.0), qv[tid], accept ? s_eff * Y[tid] : 0.0);
                }
            }
            if constexpr (FAST) fast_valid = shape_it && !dead;
            __syncthreads();
            const bool sig = FULL && a.d_sigma_trace && keep;
            double rv[NT], cv[NT];
#pragma unroll
            for (int x = 0; x < NT; ++x) {
                rv[x] = bfac * v[16 * x + ti]; // no range test: v is followed by kVPad zeros (16 NT - 1 < d + 16)
                cv[x] = v[16 * x + tk];
            }
#pragma unroll
            for (int bb = 0; bb < NT; ++bb) {
#pragma unroll
                for (int aa = 0; aa <= bb; ++aa) {
                    // C sfac + (bfac r_a) r_b as ONE instruction that overwrites C: left to the compiler the product becomes
                    // the accumulator of a v_fmac and every element is then moved back (28 v_mov_b64 per iteration at d = 100)
                    double val = Ct[blk(aa, bb)];
                    const double rr = rv[aa] * cv[bb];
                    asm("v_fma_f64 %0, %0, %1, %2" : "+v"(val) : "v"(sfac), "v"(rr));
                    Ct[blk(aa, bb)] = val;
                    if (sig) {
                        const int r = 16 * aa + ti, cl = 16 * bb + tk;
                        if (r <= cl && cl < d) {
                            a.d_sigma_trace[(t * ((int64_t)d * d) + cl * d + r) * n + c] = val;
                            if (r != cl) a.d_sigma_trace[(t * ((int64_t)d * d) + r * d + cl) * n + c] = val;
                        }
                    }
                }
            }
            // no barrier: v is next written after the first barrier of the next iteration's proposal
        }
    }

    if (tid < d) {
        if (a.d_final_theta) a.d_final_theta[(int64_t)tid * n + c] = th[tid];
        if (a.d_final_mean) a.d_final_mean[(int64_t)tid * n + c] = mn[tid];
    }
    if (a.d_final_cov) {
#pragma unroll
        for (int bb = 0; bb < NT; ++bb) {
#pragma unroll
            for (int aa = 0; aa <= bb; ++aa) {
                const int r = 16 * aa + ti, cl = 16 * bb + tk;
                if (r <= cl && cl < d) {
                    a.d_final_cov[(int64_t)(cl * d + r) * n + c] = Ct[blk(aa, bb)];
                    a.d_final_cov[(int64_t)(r * d + cl) * n + c] = Ct[blk(aa, bb)];
                }
            }
        }
    }
    if (tid == 0) {
        if (a.d_acceptance_rate) a.d_acceptance_rate[c] = rate;
        if (a.d_status) a.d_status[c] = status;
        if (a.d_shape_iterations) a.d_shape_iterations[c] = n_shape;
    }
    if (a.d_state_out) {
        double *const sp = a.d_state_out + c;
        if (tid < d) {
            sp[(int64_t)(SL.THETA + tid) * n] = th[tid];
            sp[(int64_t)(SL.MEAN + tid) * n] = mn[tid];
        }
        if constexpr (FAST) { // the factor lives in the runs; A is only kept current by the full factorizations
            if (p_col >= 0) {
#pragma unroll
                for (int u = 0; u < WS; ++u)
                    if (u < p_len) A[tri(0, p_col) + p_first + u] = W[u];
            }
            __syncthreads();
        }
        for (int e = tid; e < T; e += kBlockThreads) sp[(int64_t)(SL.R + e) * n] = A[e];
        if (tid < d) { // the last residual and its whitened form (the rank-1 mode continues from them)
            sp[(int64_t)(SL.Q + tid) * n] = FAST ? rres[tid] : (need_moments ? v[tid] : 0.0);
            sp[(int64_t)(SL.Q + d + tid) * n] = FAST ? qv[tid] : 0.0;
        }
#pragma unroll
        for (int bb = 0; bb < NT; ++bb) {
#pragma unroll
            for (int aa = 0; aa <= bb; ++aa) {
                const int r = 16 * aa + ti, cl = 16 * bb + tk;
                if (r <= cl && cl < d) sp[(int64_t)(SL.C + tri(r, cl)) * n] = Ct[blk(aa, bb)];
            }
        }
        if (tid == kBlockThreads - 64) sp[(int64_t)(SL.S + ST_ROOT) * n] = rootTerm; // the wave that keeps it (see the determinant block)
        if (tid == 0) {
            double *const ss = sp + (int64_t)SL.S * n;
            ss[ST_TAG * n] = (double)MCMC_KERNEL_BLOCK;
            ss[ST_ITER * n] = (double)(i0 + a.n_iterations);
            ss[ST_TCUR * n] = t_cur;
            ss[ST_RATE * n] = rate;
            ss[ST_SD * n] = scaling_sd;
            ss[ST_MULT * n] = scaling_multiplier;
            ss[ST_SEFF * n] = s_eff;
            ss[ST_PRODR * n] = prodR;
            ss[ST_WLO * n] = (double)(uint32_t)win.bits;
            ss[ST_WHI * n] = (double)(uint32_t)(win.bits >> 32);
            ss[ST_WLEN * n] = (double)win.len;
            ss[ST_BASE * n] = base_emp ? 1.0 : 0.0;
            ss[ST_STATUS * n] = (double)status;
            ss[ST_NSHAPE * n] = (double)n_shape;
            ss[ST_WCOUNT * n] = (double)win.count;
            ss[ST_FAST * n] = (FAST && fast_valid) ? 1.0 : 0.0;
        }
        __syncthreads(); // thread 0's last ring store is complete (the barrier drains its vector-memory counter)
        for (int wd = tid; wd < a.win_words; wd += kBlockThreads) {
            const uint64_t v = (wd == (win_pos >> 6)) ? win.bits
                                                      : (uint64_t)__hip_atomic_load(ring + (int64_t)wd * n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sp[(int64_t)(SL.TOTAL + 2 * wd) * n] = (double)(uint32_t)v;
            sp[(int64_t)(SL.TOTAL + 2 * wd + 1) * n] = (double)(uint32_t)(v >> 32);
        }
    }
}

#ifdef MCMC_USER_BLOCK
// log-density of the user target at n points, x as [d][n]: one workgroup per point (mcmc_target_eval_batch, d > 12)
__global__ __launch_bounds__(kBlockThreads) void block_target_eval_kernel(int64_t n, int d, const double *__restrict__ x,
                                                                           const double *__restrict__ uprm, double *__restrict__ out)
{
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const int64_t c = blockIdx.x;
    if (tid < d) lds[tid] = x[(int64_t)tid * n + c];
    __syncthreads();
    double acc = ::mcmc_user_target_part(lds, d, uprm, tid, kBlockThreads);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((tid & 63) == 0) lds[d + (tid >> 6)] = acc;
    __syncthreads();
    if (tid == 0) out[c] = (lds[d] + lds[d + 1]) + (lds[d + 2] + lds[d + 3]);
}
#endif

#ifndef __HIPCC_RTC__
// user == false: the Gaussian target (p.tparams = m[d], H[d*d]); user == true: p.tparams are the user's parameters
int launch_rwm_block_with(const RwmArgs &a, const RwmParams &p, hipStream_t s, bool user, const RwmLauncher &launch)
{
    const int d = a.d;
    if (d < 1 || d > kMaxDimRwm) return set_error(MCMC_E_UNSUPPORTED, "adaptive RWM: d=%d outside 1..%d", d, kMaxDimRwm);
    if (!user && (int)p.tparams.size() != d + d * d) return set_error(MCMC_E_BAD_ARG, "gaussian target needs d + d*d parameters");
    const BlockLayout L(d);
    const BlockSegments tseg = user ? BlockSegments{} : block_segments(d, 0, 8), pseg = block_segments(d, 1, 4);
    const int TP = tseg.S > 0 ? tseg.S : 1;
    if (TP + 7 > BlockLayout::kVPad + 1) return set_error(MCMC_E_UNSUPPORTED, "adaptive RWM: d=%d needs a longer zero pad", d);
    std::vector<double> blob(user ? (size_t)L.hq() + p.tparams.size() + 1 : (size_t)L.total(TP), 0.0);
    for (int k = 0; k < d; ++k) blob[L.init() + k] = p.init[k];
    for (int e = 0; e < L.T; ++e) {
        blob[L.cov() + e] = p.cov_packed[e];
        blob[L.r0() + e] = p.R0_packed[e];
    }
    blob[L.root0()] = 1.0 / std::sqrt(a.twopi_pow_d * p.det0);
    const double *Hm = user ? nullptr : p.tparams.data() + d; // column-major precision matrix
    for (int k = 0; k < d && !user; ++k) {
        blob[L.tmean() + k] = p.tparams[k];
        blob[L.hdiag() + k] = Hm[(size_t)k * d + k];
    }
    for (size_t e = 0; e < p.tparams.size() && user; ++e) blob[L.hq() + e] = p.tparams[e];
    blob[L.meta()] = TP;
    // rows of Hq in LDS: what is left of a CU's 160 KB beside the state of the chains it holds (launch bounds: 4 / 3 / 2)
    const int nt_ = (d + 15) / 16, per_cu = nt_ <= 2 ? 4 : (nt_ <= 4 ? 3 : 2);
    const bool fast = a.factor_rank1 != 0;
    if (fast && user) return set_error(MCMC_E_UNSUPPORTED, "MCMC_FACTOR_RANK1 serves the Gaussian target only");
    const long own_doubles = L.lds_doubles() + (fast ? L.fast_doubles() : 0);
    const long free_doubles = (160 * 1024 / per_cu - 1024) / 8 - own_doubles;
    const int KL = (user || nt_ < 3 || nt_ > 7) ? 0 : (int)std::max(0L, std::min<long>(TP, free_doubles / kBlockThreads)) & ~3;
    blob[L.meta() + 1] = KL;
    for (int tid = 0; tid < kBlockThreads; ++tid) {
        const bool ht = tid < (int)tseg.row.size(), hp = tid < (int)pseg.row.size();
        blob[L.ttab() + 2 * tid] = ht ? tseg.row[tid] : -1;
        blob[L.ttab() + 2 * tid + 1] = ht ? tseg.first[tid] : 0;
        blob[L.ptab() + 3 * tid] = hp ? pseg.row[tid] : -1;
        blob[L.ptab() + 3 * tid + 1] = hp ? pseg.first[tid] : 0;
        blob[L.ptab() + 3 * tid + 2] = hp ? pseg.count[tid] : 0;
        if (ht)
            for (int t = 0; t < tseg.count[tid]; ++t) {
                const int row = tseg.row[tid], i = tseg.first[tid] + t;
                blob[L.hq() + (size_t)t * kBlockThreads + tid] = Hm[(size_t)row * d + i] + Hm[(size_t)i * d + row];
            }
    }
    for (int r = 0; r < (int)pseg.row.size(); ++r) { // runs of a column are consecutive threads
        const int cl = pseg.row[r];
        if (blob[L.pred() + 2 * cl + 1] == 0) blob[L.pred() + 2 * cl] = r;
        blob[L.pred() + 2 * cl + 1] += 1;
    }
    double *d_prm = nullptr;
    MCMC_HIP_CHECK(hipMallocAsync((void **)&d_prm, blob.size() * sizeof(double), s));
    if (const int ru = upload_block(d_prm, blob.data(), blob.size(), s); ru != MCMC_OK) {
        (void)hipFreeAsync(d_prm, s);
        return ru;
    }
    const size_t ldsb = ((size_t)own_doubles + (size_t)KL * kBlockThreads) * sizeof(double);
    const bool full = a.d_Z || a.d_sigma_trace || a.d_accepted || a.d_log_acceptance || a.d_log_A || a.d_scaling_sd ||
                      a.d_scaling_multiplier || a.d_rate_trace;
    hipError_t e1 = launch(fast ? 2 : (full ? 1 : 0), (unsigned)a.n_chains, (unsigned)kBlockThreads, ldsb, s, a, d_prm);
    hipError_t e2 = hipFreeAsync(d_prm, s);
    if (e1 != hipSuccess) return set_error(MCMC_E_HIP, "rwm_block_kernel launch failed: %s", hipGetErrorString(e1));
    if (e2 != hipSuccess) return set_error(MCMC_E_HIP, "hipFreeAsync failed: %s", hipGetErrorString(e2));
    return MCMC_OK;
}

int launch_rwm_block(const RwmArgs &a, const RwmParams &p, hipStream_t s)
{
    if (p.target_kind == MCMC_TARGET_USER) return launch_rwm_block_user(a, p, s);
    if (p.target_kind != MCMC_TARGET_GAUSSIAN_PREC)
        return set_error(MCMC_E_BAD_ARG, "target kind %d is 2-dimensional; d=%d", p.target_kind, a.d);
    const int nt = (a.d + 15) / 16;
    return launch_rwm_block_with(
        a, p, s, false,
        [nt](int full, unsigned grid, unsigned block, size_t ldsb, hipStream_t st, const RwmArgs &aa, const double *d_prm) {
            auto go = [&](auto kern) {
                hipError_t e0 = ensure_dynamic_lds((const void *)kern, ldsb);
                if (e0 != hipSuccess) return e0;
                hipLaunchKernelGGL(kern, dim3(grid), dim3(block), ldsb, st, aa, d_prm);
                return hipGetLastError();
            };
            switch (nt) {
#define MCMC_CASE(NT)                                                                  \
    case NT:                                                                           \
        return full == 2 ? go(rwm_block_kernel<NT, false, false, true>)                \
                         : (full ? go(rwm_block_kernel<NT, true, false>) : go(rwm_block_kernel<NT, false, false>));
                MCMC_CASE(1) MCMC_CASE(2) MCMC_CASE(3) MCMC_CASE(4) MCMC_CASE(5) MCMC_CASE(6) MCMC_CASE(7) MCMC_CASE(8)
#undef MCMC_CASE
            default: return hipErrorInvalidValue;
            }
        });
}
#endif // !__HIPCC_RTC__

} // namespace mcmc
