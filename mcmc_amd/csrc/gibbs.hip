// Truncated multivariate normal Gibbs sweep on gfx950.
//
// Replaces the sweep loop of rtmvnorm_gibbs (tmvrnormGibbs.cpp:115-139) and the draw loop of
// rtnorm_gibbs (:58-67).  One chain per lane: a wavefront advances 64 independent chains in
// lockstep, so the 2 x pnorm + qnorm work of every coordinate (about 70 % of a sweep) runs on
// all 64 lanes; a wavefront-per-chain layout would leave 63 of 64 lanes idle there.
// Coordinates inside a sweep are visited in order (Gauss-Seidel), as the reference does.
//
// Data layout
//   * chain state: the centred vector x - mu in LDS as [k][lane] (d * 512 B per wavefront;
//     ds_read_b64 / ds_write_b64 with bank = lane, conflict free).  d = 10 -> 5 KiB per wave, so
//     eight waves per SIMD fit in the 160 KiB of a CU; d = 128 -> 64 KiB, two waves per CU.
//   * shared parameters (mu, sd, bounds, regression rows P_i): one block in HBM, read with
//     wave-uniform addresses, i.e. through the scalar cache into SGPRs (s_load_dwordx*), and
//     used directly as the scalar operand of v_fma_f64.  The offset `z` below is an opaque zero
//     refreshed every sweep: it keeps the compiler from hoisting ~200 parameter SGPRs out of
//     the loop and then spilling them to VGPR lanes (v_readlane in the hot loop).
//   * samples: one coalesced store per coordinate and kept sweep, trace[(t*d+k)*n + c] --
//     512 contiguous bytes per wavefront.
#include <mutex>

#include <map>
#include <mutex>
#include <utility>

#include "common.h"
#include "normal_cdf.h"
#include "philox.h"

namespace mcmc {

// parameter block: d rows of W = d + 4 doubles -- P_i (d-1 entries), mu_i, sd_i, 1/sd_i, lower_i, upper_i: everything
// coordinate i needs behind ONE scalar pointer -- then mu[d], init[d]
template <bool TAIL_SAFE, bool MOMENTS>
__global__ __launch_bounds__(64, 3) void gibbs_kernel(const GibbsArgs a, const double *__restrict__ prm)
{
    extern __shared__ double xc[]; // centred state, [k][lane]
    const int lane = threadIdx.x;
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    const int64_t n = a.n_chains;
    const bool live = c < n;
    const int64_t cc = live ? c : n - 1; // tail lanes shadow the last chain and store nothing
    const int d = a.d;
    const uint64_t gchain = (uint64_t)(a.chain_offset + cc);
    const PhiloxKey key(a.seed);
    const bool inject = a.d_U != nullptr;

    for (int k = 0; k < d; ++k) {
        const double x0 = a.d_init ? a.d_init[(int64_t)k * n + cc] : prm[(d + 4) * d + d + k];
        xc[k * 64 + lane] = x0 - prm[(d + 4) * d + k];
        if (MOMENTS && live) {
            a.d_mean[(int64_t)k * n + c] = 0.0;
            a.d_m2[(int64_t)k * n + c] = 0.0;
        }
    }
    int32_t status = MCMC_CHAIN_OK;

    for (int j = 1; j <= a.n_sweeps; ++j) {
        int z = 0;
        asm volatile("" : "+s"(z)); // opaque zero: parameter loads stay inside the sweep loop
        const double *__restrict__ rows = prm + z;
        const int ja = a.sweep_offset + j; // absolute sweep: Philox counter, thinning, status (resume)
        const bool keep = a.d_trace && (ja % a.thin) == 0;
        const int64_t t = ja / a.thin - a.sweep_offset / a.thin - 1;
        const bool mom = MOMENTS && j > a.burn;
        const double inv = mom ? 1.0 / (double)(j - a.burn) : 0.0;
        double ua = 0.0, ub = 0.0;
        for (int i = 0; i < d; ++i) {
            double u;
            if (inject) {
                u = a.d_U[((int64_t)(j - 1) * d + i) * n + cc];
            } else {
                if ((i & 1) == 0) {
                    // opaque copy of the key: the ten round keys are then formed by scalar adds here; hoisted out of the
                    // loops they take twenty scalar registers, which spill to VGPR lanes and come back through v_readlane
                    PhiloxKey kk = key;
                    asm volatile("" : "+s"(kk.k0), "+s"(kk.k1));
                    philox_uniform2(kk, gchain, (uint32_t)ja, TAG_GIBBS, (uint32_t)(i >> 1), ua, ub);
                }
                u = (i & 1) ? ub : ua;
            }
            // conditional mean mu_i + P_i (x[-i] - mu[-i])   (tmvrnormGibbs.cpp:124-127)
            const double *__restrict__ Pi = rows + (int64_t)i * (d + 4);
            double s = 0.0;
            for (int k = 0; k < i; ++k) s += Pi[k] * xc[k * 64 + lane];
            for (int k = i + 1; k < d; ++k) s += Pi[k - 1] * xc[k * 64 + lane];
            const double mu0 = Pi[d - 1];
            const double mu_i = mu0 + s;
            const double xi = rtnorm_draw_std<TAIL_SAFE>(u, mu_i, Pi[d], Pi[d + 1], Pi[d + 2], Pi[d + 3]);
            xc[i * 64 + lane] = xi - mu0;
            if (!isfinite(xi) && status == MCMC_CHAIN_OK) status = (ja << 4) | MCMC_CHAIN_NONFINITE_STATE;
            if (live) {
                if (keep) a.d_trace[(t * d + i) * n + c] = xi;
                if (j == a.n_sweeps && a.d_final) a.d_final[(int64_t)i * n + c] = xi;
                if (mom) {
                    const int64_t o = (int64_t)i * n + c;
                    double mean = a.d_mean[o];
                    const double delta = xi - mean;
                    mean += delta * inv;
                    a.d_mean[o] = mean;
                    a.d_m2[o] += delta * (xi - mean);
                }
            }
        }
    }
    if (live && a.d_status) a.d_status[c] = status;
}

// ------------------------------------------------------------------------------------------
// Wide variant for d > 27.  There the regression rows P no longer fit the 16 KB scalar cache (130 KB at
// d = 128) and an LDS-resident state (64 KB per 64 chains at d = 128) would leave room for two chain groups
// per CU: with the kernel above that is two waves on four SIMDs, each exposed to every scalar-load and
// LDS latency (measured 34 000 cycles per coordinate and wave, 16 times the d = 10 figure).
//
// Here a workgroup holds G chain groups.  Per group a DRAW wave and NDW DOT waves work one coordinate apart:
//   * while the draw wave turns coordinate i into a sample (2 x pnorm, qnorm: the long dependent
//     chain), the dot waves form for coordinate i+1 the conditional-mean sum over every k except i and
//     i+1 -- x_i is the only state element that changes meanwhile -- and the uniform u_(i+1), and
//     leave both in a double-buffered LDS slot;
//   * the draw wave then adds the one missing term P[i+1][i] x_i (x_i is still in a register).
// So the d-term dot product is off the critical path.
// The rows are shared by the workgroup: every thread fetches one or two values of row i+2 (P and the
// five scalars, coalesced) while step i runs and drops them into a three-slot LDS ring before the
// single barrier that ends the step; the barrier waits for LDS traffic only, so trace and moment
// stores stay in flight.  Running moments are kept as sums of the shifted draw and its square, added to
// HBM with no-return atomics (each lane owns its addresses) and turned into mean / M2 once at the end.
// The sum is associated differently from the reference's left-to-right loop
// (tmvrnormGibbs.cpp:124-127): the x_(i-1) term is added last.
//
// rows block in HBM: d rows of W doubles: P_i padded to d entries (entry i is 0 and never read), mu_i,
// sd_i, 1/sd_i, lower_i, upper_i; then init[d].
constexpr int kWideMaxPrefetch = 3; // ceil(W / threads) at d = 256 (two groups of five waves, 640 threads: 1)

// The state lives in REGISTERS (one dot wave per group up to d = 64, two up to 128, four above): with the centred
// state of 64 chains in LDS (d x 512 B: 64 KB at d = 128) a CU holds two chain groups and the draw wave's dependent
// chain (~4 400 cycles per coordinate) is hidden only twice over.  Here each dot wave of a group keeps ITS 64
// coordinates of the state in registers (static indices only: the dot product is 64 unrolled FMAs against broadcast
// LDS reads of the row, one LDS read per FMA; a new x_i arrives through a 512-byte LDS slot and is put in place by a
// select sweep over one eight-register block), the LDS holds only the row ring and the hand-over slots, and a
// workgroup runs four groups = twelve waves.
//
// Ring row (kRegTail + 64 NDW doubles): entries [0, 64 NDW) are what the dot waves multiply their registers with --
// P_i with the two entries they must not use (the coordinate itself and the one being drawn) zeroed, and ZERO from d
// on, so that a 64-wide window never meets anything but P entries and zeros (the bounds may be +-Inf, and Inf * 0 is
// NaN); the scalars of the coordinate sit behind the window: mu_i, sd_i, 1/sd_i, lower_i, upper_i, the moment shift
// and P_i[prev(i)] for the draw wave.
constexpr int kRegOwn = 64;
constexpr int kRegTail = 8; // mu, sd, 1/sd, lower, upper, shift, P[prev], pad
constexpr int kRowScalars = 6; // scalars per coordinate in the HBM rows block: mu, sd, 1/sd, lower, upper, shift
template <int Q>
struct DotPos {
    static constexpr int value = Q;
};

// dot waves per group NDW (1: d <= 64, 2: d <= 128, 4: d <= 256) and with it the group count are compile-time constants
__host__ __device__ constexpr int reg_groups(int ndw) { return ndw == 1 ? 6 : (ndw == 2 ? 4 : 2); } // 12 / 12 / 10 waves, three per SIMD at most

template <bool TAIL_SAFE, bool MOMENTS, int NDW>
__global__ __launch_bounds__(768) void gibbs_reg_kernel(const GibbsArgs a, const double *__restrict__ rows, int W)
{
    constexpr int G = reg_groups(NDW);
    extern __shared__ double lds[]; // [3][64 NDW + kRegTail] row ring, [G][2][NDW + 1][64] slots (partial sums, u), [G][2][64] new x
    constexpr int NS = NDW + 1;
    constexpr int nthreads = G * (1 + NDW) * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool dotter = wave >= G;
    const int group = dotter ? (wave - G) % G : wave;
    const int part = dotter ? (wave - G) / G : 0;
    const int d = a.d;
    constexpr int TW = NDW * kRegOwn;  // dot-visible entries of a ring row
    constexpr int W2 = TW + kRegTail;  // ring row stride
    const int64_t c = ((int64_t)blockIdx.x * G + group) * 64 + lane;
    const int64_t n = a.n_chains;
    const bool live = c < n;
    const int64_t cc = live ? c : n - 1; // lanes past the end shadow the last chain and store nothing
    double *const ring = lds;
    double *const slot = lds + 3 * W2 + (size_t)group * (2 * NS * 64) + lane;               // [parity][sums.., u][lane]
    double *const xnew = lds + 3 * W2 + (size_t)G * (2 * NS * 64) + (size_t)group * 128 + lane; // [parity][lane]
    const double *const init = rows + (size_t)d * W;
    const int64_t total = (int64_t)a.n_sweeps * d;
    const int base = part * kRegOwn; // first coordinate this dot wave owns

    // ring row for coordinate i: P_i with the entries i (zero already) and prev(i) zeroed, P_i[prev(i)] and the
    // scalars behind the dot waves' window; entries d .. TW-1 keep the zeros written here once
    auto put_row = [&](double *dst, int i, int e, double val) {
        const int prev = (i == 0) ? d - 1 : i - 1;
        if (e >= d) {
            if (e < d + kRowScalars) dst[TW + (e - d)] = val;
            return;
        }
        if (e == prev) {
            dst[TW + kRowScalars] = val;
            val = 0.0;
        }
        dst[e] = val;
    };
    for (int e = tid; e < 3 * W2; e += nthreads) ring[e] = 0.0;
    __syncthreads();
    for (int e = tid; e < 2 * W; e += nthreads) {
        const int r = e / W, k = e - r * W;
        put_row(ring + r * W2, r, k, rows[e]);
    }

    if (!dotter && MOMENTS && live) {
        for (int k = 0; k < d; ++k) {
            a.d_mean[(int64_t)k * n + c] = 0.0;
            a.d_m2[(int64_t)k * n + c] = 0.0;
        }
    }
    __syncthreads();

    // the ring bookkeeping every wave does in every step: fetch row i+2, and after the step's work put it in place
    int i = 0, j = 1, r0 = 0; // coordinate, sweep, ring slot of row i
    auto fetch = [&](int i2, double (&pre)[kWideMaxPrefetch]) {
#pragma unroll
        for (int q = 0; q < kWideMaxPrefetch; ++q) {
            const int e = tid + q * nthreads;
            pre[q] = (e < W) ? rows[(size_t)i2 * W + e] : 0.0;
        }
    };
    auto store = [&](int i2, int r2, const double (&pre)[kWideMaxPrefetch]) {
        double *const nxt = ring + r2 * W2;
#pragma unroll
        for (int q = 0; q < kWideMaxPrefetch; ++q) {
            const int e = tid + q * nthreads;
            if (e < W) put_row(nxt, i2, e, pre[q]);
        }
    };

    // ================================ dot waves =====================================================
    // (a loop of their own: the 128 registers of state must not be live in the draw wave's code)
    if (dotter) {
        double x[kRegOwn]; // centred state of coordinates base .. base + 63 (zero beyond d)
#pragma unroll
        for (int k = 0; k < kRegOwn; ++k) {
            const int g = base + k;
            x[k] = 0.0;
            if (g < d) x[k] = (a.d_init ? a.d_init[(int64_t)g * n + cc] : init[g]) - rows[(size_t)g * W + d];
        }
        double ua = 0.0, ub = 0.0; // the two uniforms of the current Philox block (first dot wave)
        auto prepare = [&](int ip, int jp, const double *row, int p) {
            const double *const rp = row + base; // base + 63 < TW; entries from d on are zero, and so is x there
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k = 0; k < kRegOwn; k += 2) {
                s0 = fma(rp[k], x[k], s0);
                s1 = fma(rp[k + 1], x[k + 1], s1);
            }
            slot[(p * NS + part) * 64] = s0 + s1;
            if (part != 0) return;
            double u;
            if (a.d_U) {
                u = a.d_U[((int64_t)(jp - 1) * d + ip) * n + cc];
            } else {
                if ((ip & 1) == 0)
                    philox_uniform2(PhiloxKey(a.seed), (uint64_t)(a.chain_offset + cc), (uint32_t)(a.sweep_offset + jp), TAG_GIBBS, (uint32_t)(ip >> 1), ua, ub);
                u = (ip & 1) ? ub : ua;
            }
            slot[(p * NS + NDW) * 64] = u;
        };
        prepare(0, 1, ring, 0);
        lds_barrier();
        // One step: fetch row i+2, put the x drawn in the previous step into its register, prepare coordinate i+1.
        // The coordinate loop is unrolled by eight so that the position of that x inside its block of eight registers
        // is a compile-time constant QP: eight selects (one per block) instead of a sweep over all 64 registers.
        int64_t t = 0;
        auto step = [&](auto qc) {
            constexpr int Q = decltype(qc)::value;
            const int r1 = (r0 == 2) ? 0 : r0 + 1, r2 = (r1 == 2) ? 0 : r1 + 1;
            const int i1 = (i + 1 == d) ? 0 : i + 1, i2 = (i1 + 1 == d) ? 0 : i1 + 1;
            const int p = (int)(t & 1);
            double pre[kWideMaxPrefetch];
            fetch(i2, pre);
            if (t > 0) {
                const int il = ((i == 0) ? d - 1 : i - 1) - base;
                if (il >= 0 && il < kRegOwn) {
                    const double v = xnew[(p ^ 1) * 64];
                    if (i == 0) { // the wrap: position (d - 1) % 8 is not known at compile time
#pragma unroll
                        for (int k = 0; k < kRegOwn; ++k) x[k] = (k == il) ? v : x[k];
                    } else {
                        constexpr int QP = (Q + 7) & 7; // (i - 1) % 8
                        const int B = il >> 3;
#pragma unroll
                        for (int bq = 0; bq < kRegOwn / 8; ++bq) x[8 * bq + QP] = (bq == B) ? v : x[8 * bq + QP];
                    }
                }
            }
            if (t + 1 < total) prepare(i1, i1 == 0 ? j + 1 : j, ring + r1 * W2, p ^ 1);
            store(i2, r2, pre);
            lds_barrier();
            r0 = r1;
            i = i1;
            if (i == 0) ++j;
            ++t;
        };
        for (int js = 0; js < a.n_sweeps; ++js) {
            for (int ib = 0; ib < d; ib += 8) { // i == ib + Q inside the unrolled body
                step(DotPos<0>{});
                if (ib + 1 < d) step(DotPos<1>{});
                if (ib + 2 < d) step(DotPos<2>{});
                if (ib + 3 < d) step(DotPos<3>{});
                if (ib + 4 < d) step(DotPos<4>{});
                if (ib + 5 < d) step(DotPos<5>{});
                if (ib + 6 < d) step(DotPos<6>{});
                if (ib + 7 < d) step(DotPos<7>{});
            }
        }
        return;
    }

    // ================================ draw wave =====================================================
    int32_t status = MCMC_CHAIN_OK;
    double xprev = (a.d_init ? a.d_init[(int64_t)(d - 1) * n + cc] : init[d - 1]) - rows[(size_t)(d - 1) * W + d]; // centred x of the previous coordinate
    lds_barrier();
    for (int64_t t = 0; t < total; ++t) {
        const int r1 = (r0 == 2) ? 0 : r0 + 1, r2 = (r1 == 2) ? 0 : r1 + 1;
        const int i1 = (i + 1 == d) ? 0 : i + 1, i2 = (i1 + 1 == d) ? 0 : i1 + 1;
        const int p = (int)(t & 1);
        double pre[kWideMaxPrefetch];
        fetch(i2, pre);
        {
            const double *const row = ring + r0 * W2;
            const int ja = a.sweep_offset + j; // absolute sweep (resume)
            const bool keep = a.d_trace && (ja % a.thin) == 0;
            const bool mom = MOMENTS && j > a.burn;
            // conditional mean mu_i + P_i (x[-i] - mu[-i])   (tmvrnormGibbs.cpp:124-127)
            double s = slot[(p * NS) * 64];
            for (int q = 1; q < NDW; ++q) s += slot[(p * NS + q) * 64];
            const double *const sc = row + TW; // mu, sd, 1/sd, lower, upper, shift, P_i[prev]
            s += sc[kRowScalars] * xprev;
            const double u = slot[(p * NS + NDW) * 64];
            const double mui = sc[0];
            const double xi = rtnorm_draw_std<TAIL_SAFE>(u, mui + s, sc[1], sc[2], sc[3], sc[4]);
            xprev = xi - mui;
            xnew[p * 64] = xprev;
            if (!isfinite(xi) && status == MCMC_CHAIN_OK) status = (ja << 4) | MCMC_CHAIN_NONFINITE_STATE;
            if (live) {
                if (keep) a.d_trace[((int64_t)(ja / a.thin - a.sweep_offset / a.thin - 1) * d + i) * n + c] = xi;
                if (j == a.n_sweeps && a.d_final) a.d_final[(int64_t)i * n + c] = xi;
                if (mom) { // running sums of x - shift_i and its square; no value comes back, so nothing waits for HBM
                    const double xs = xi - sc[5];
                    unsafeAtomicAdd(a.d_mean + (int64_t)i * n + c, xs);
                    unsafeAtomicAdd(a.d_m2 + (int64_t)i * n + c, xs * xs);
                }
            }
        }
        store(i2, r2, pre);
        lds_barrier();
        r0 = r1;
        i = i1;
        if (i == 0) ++j;
    }
    if (live) {
        if (a.d_status) a.d_status[c] = status;
        if (MOMENTS) { // sums of (x - shift_k), (x - shift_k)^2 over the counted sweeps -> mean and M2 (sum of squared deviations).
            // shift_k is the point of [lower_k, upper_k] nearest mu_k (launch_gibbs): within the spread of the draws
            // of the data in the tail cases too, so the difference below cancels few digits
            __threadfence();
            const int nm = a.n_sweeps - a.burn;
            const double inv = nm > 0 ? 1.0 / (double)nm : 0.0;
            for (int k = 0; k < d; ++k) {
                const double s1 = a.d_mean[(int64_t)k * n + c], s2 = a.d_m2[(int64_t)k * n + c];
                a.d_mean[(int64_t)k * n + c] = rows[(size_t)k * W + d + 5] + s1 * inv;
                a.d_m2[(int64_t)k * n + c] = fma(-s1 * inv, s1, s2);
            }
        }
    }
}

// uniforms a Philox Gibbs run draws, [n_sweeps][d][n_chains]
__global__ __launch_bounds__(256) void gibbs_streams_kernel(uint64_t seed, int64_t chain_offset, int64_t n,
                                                            int32_t n_sweeps, int32_t d, double *U)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const PhiloxKey key(seed);
    const uint64_t gchain = (uint64_t)(chain_offset + c);
    for (int j = 1; j <= n_sweeps; ++j) {
        for (int b = 0; 2 * b < d; ++b) {
            double u0, u1;
            philox_uniform2(key, gchain, (uint32_t)j, TAG_GIBBS, (uint32_t)b, u0, u1);
            U[((int64_t)(j - 1) * d + 2 * b) * n + c] = u0;
            if (2 * b + 1 < d) U[((int64_t)(j - 1) * d + 2 * b + 1) * n + c] = u1;
        }
    }
}

// rtnorm_gibbs (tmvrnormGibbs.cpp:49-70): Fa, Fb once (:55-56), then one quantile per draw.
template <bool TAIL_SAFE>
__global__ __launch_bounds__(256) void rtnorm_kernel(int64_t n, double mu, double sigma, double lo, double hi,
                                                     const double *F, uint64_t seed, int64_t offset, double *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double u;
    if (F) {
        u = F[i];
    } else {
        const uint64_t g = (uint64_t)(offset + i);
        double u0, u1;
        philox_uniform2(PhiloxKey(seed), g >> 1, 0u, TAG_RTNORM, 0u, u0, u1);
        u = (g & 1) ? u1 : u0;
    }
    if (TAIL_SAFE) {
        out[i] = rtnorm_draw<true>(u, mu, sigma, lo, hi);
    } else {
        const double Fa = pnorm(lo, mu, sigma);
        const double Fb = pnorm(hi, mu, sigma);
        const double p_i = u * (Fb - Fa) + Fa;
        out[i] = mu + sigma * qnorm_std(p_i);
    }
}

// ------------------------------------------------------------------------------------------
// Shared parameter block of a Gibbs launch: the lane kernel's blob (d <= kGibbsLaneMaxDim) or the wide kernel's rows.
static size_t gibbs_block_doubles(int d)
{
    if (d > kGibbsLaneMaxDim) return (size_t)d * ((d + kRowScalars + 1) & ~1) + d;
    return (size_t)(d + 4) * d + 2 * (size_t)d;
}

static void build_gibbs_block(const GibbsParams &p, double *blk)
{
    const int d = p.d;
    const size_t total = gibbs_block_doubles(d);
    for (size_t e = 0; e < total; ++e) blk[e] = 0.0;
    if (d > kGibbsLaneMaxDim) { // wide variant: rows of W doubles, shared by the waves of a workgroup
        const int W = (d + kRowScalars + 1) & ~1;
        for (int i = 0; i < d; ++i) {
            double *r = blk + (size_t)i * W;
            for (int k = 0, cidx = 0; k < d; ++k)
                if (k != i) r[k] = p.P[(size_t)i * (d - 1) + cidx++];
            r[d] = p.mu[i];
            r[d + 1] = p.sd[i];
            r[d + 2] = 1.0 / p.sd[i];
            r[d + 3] = p.lower[i];
            r[d + 4] = p.upper[i];
            r[d + 5] = std::fmin(std::fmax(p.mu[i], p.lower[i]), p.upper[i]); // moment shift: the point of the box nearest mu (finite)
        }
        for (int k = 0; k < d; ++k) blk[(size_t)d * W + k] = p.init[k];
        return;
    }
    for (int i = 0; i < d; ++i) {
        double *r = blk + (size_t)i * (d + 4);
        for (int k = 0; k < d - 1; ++k) r[k] = p.P[(size_t)i * (d - 1) + k];
        r[d - 1] = p.mu[i];
        r[d] = p.sd[i];
        r[d + 1] = 1.0 / p.sd[i];
        r[d + 2] = p.lower[i];
        r[d + 3] = p.upper[i];
        blk[(size_t)(d + 4) * d + i] = p.mu[i];
        blk[(size_t)(d + 4) * d + d + i] = p.init[i];
    }
}

// the launch proper, parameter block already on the device (and ordered before this launch on `s`)
static int launch_gibbs_on_block(const GibbsArgs &a, bool tail_safe, const double *d_blk, hipStream_t s)
{
    const int d = a.d;
    const bool moments = a.d_mean != nullptr;
    if (d > kGibbsLaneMaxDim) {
        const int W = (d + kRowScalars + 1) & ~1;
        // state in the dot waves' registers, 64 coordinates each; group and wave counts are functions of d only, so
        // that results do not depend on how chains are split over launches
        const int NDW = d <= kRegOwn ? 1 : (d <= 2 * kRegOwn ? 2 : 4); // dot waves per group
        const int G = reg_groups(NDW);                                 // groups per workgroup
        const int threads = G * (1 + NDW) * 64;
        if ((W + threads - 1) / threads > kWideMaxPrefetch) return set_error(MCMC_E_UNSUPPORTED, "gibbs: d=%d too wide", d);
        const size_t ldsr = ((size_t)3 * (NDW * kRegOwn + kRegTail) + (size_t)G * (2 * (NDW + 1) * 64 + 128)) * sizeof(double);
        const int64_t gridr = (a.n_chains + 64 * G - 1) / (64 * G);
        auto gor = [&](auto kern) -> int {
            hipLaunchKernelGGL(kern, dim3((unsigned)gridr), dim3(threads), ldsr, s, a, d_blk, W);
            hipError_t el = hipGetLastError();
            return el == hipSuccess ? MCMC_OK : set_error(MCMC_E_HIP, "gibbs_reg_kernel launch failed: %s", hipGetErrorString(el));
        };
        auto pick = [&](auto ndw) -> int {
            constexpr int N = decltype(ndw)::value;
            if (tail_safe) return moments ? gor(gibbs_reg_kernel<true, true, N>) : gor(gibbs_reg_kernel<true, false, N>);
            return moments ? gor(gibbs_reg_kernel<false, true, N>) : gor(gibbs_reg_kernel<false, false, N>);
        };
        return NDW == 1 ? pick(DotPos<1>{}) : (NDW == 2 ? pick(DotPos<2>{}) : pick(DotPos<4>{}));
    }
    const size_t lds = (size_t)d * 64 * sizeof(double);
    const int64_t grid = (a.n_chains + 63) / 64;
    auto go = [&](auto kern) -> int {
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds, s, a, d_blk);
        hipError_t el = hipGetLastError();
        return el == hipSuccess ? MCMC_OK : set_error(MCMC_E_HIP, "gibbs_kernel launch failed: %s", hipGetErrorString(el));
    };
    if (tail_safe) return moments ? go(gibbs_kernel<true, true>) : go(gibbs_kernel<true, false>);
    return moments ? go(gibbs_kernel<false, true>) : go(gibbs_kernel<false, false>);
}

hipError_t ensure_dynamic_lds(const void *kern, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> have; // (kernel, device) -> largest size set so far
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t &h = have[{kern, dev}];
    if (bytes <= h) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) h = bytes;
    return e;
}

// Host doubles -> device memory, ordered on `s`, WITHOUT waiting for the stream: the data travels in the ARGUMENTS of
// small fill kernels (copied when the launch is made), so the host buffer may be a local of the caller and nothing
// queued on the stream earlier is waited for.  Above kUploadByArgs doubles (64 launches): one copy and one wait.
constexpr int kFillChunk = 448; // 3 584 B of kernel arguments per fill launch
constexpr size_t kUploadByArgs = 64 * (size_t)kFillChunk;
struct FillChunk {
    double v[kFillChunk];
};
__global__ __launch_bounds__(kFillChunk) void scratch_fill_kernel(double *__restrict__ dst, int count, const FillChunk c)
{
    const int t = threadIdx.x;
    if (t < count) dst[t] = c.v[t];
}
int upload_block(double *d_dst, const double *h_src, size_t count, hipStream_t s)
{
    static_assert(sizeof(FillChunk) + 16 <= 4096, "fill arguments must stay inside 4 KB");
    if (count > kUploadByArgs) {
        hipError_t e = hipMemcpyAsync(d_dst, h_src, count * sizeof(double), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s); // the host buffer may not outlive the call
        if (e != hipSuccess) return set_error(MCMC_E_HIP, "parameter upload failed: %s", hipGetErrorString(e));
        return MCMC_OK;
    }
    for (size_t off = 0; off < count; off += kFillChunk) {
        const int cnt = (int)(count - off < (size_t)kFillChunk ? count - off : (size_t)kFillChunk);
        FillChunk c;
        for (int t = 0; t < cnt; ++t) c.v[t] = h_src[off + t];
        for (int t = cnt; t < kFillChunk; ++t) c.v[t] = 0.0;
        hipLaunchKernelGGL(scratch_fill_kernel, dim3(1), dim3(kFillChunk), 0, s, d_dst + off, cnt, c);
        const hipError_t el = hipGetLastError();
        if (el != hipSuccess) return set_error(MCMC_E_HIP, "scratch_fill_kernel launch failed: %s", hipGetErrorString(el));
    }
    return MCMC_OK;
}

int launch_gibbs(const GibbsArgs &a, const GibbsParams &p, bool tail_safe, hipStream_t s)
{
    const int d = a.d;
    if (d < 1 || d > kMaxDimGibbs) return set_error(MCMC_E_UNSUPPORTED, "gibbs: d=%d outside 1..%d", d, kMaxDimGibbs);
    std::vector<double> blk(gibbs_block_doubles(d));
    build_gibbs_block(p, blk.data());
    double *d_blk = nullptr;
    MCMC_HIP_CHECK(hipMallocAsync((void **)&d_blk, blk.size() * sizeof(double), s));
    int rc = upload_block(d_blk, blk.data(), blk.size(), s);
    if (rc == MCMC_OK) rc = launch_gibbs_on_block(a, tail_safe, d_blk, s);
    hipError_t e = hipFreeAsync(d_blk, s);
    if (rc == MCMC_OK && e != hipSuccess) rc = set_error(MCMC_E_HIP, "hipFreeAsync failed: %s", hipGetErrorString(e));
    return rc;
}

// ------------------------------------------------------------------------------------------
// The no-allocation variant (mcmc_tmvn_gibbs_batch_noalloc): the parameter block lives in a file-scope device
// array instead of a hipMallocAsync'ed buffer.  `g_gibbs_scratch` is a HOST-side stand-in for that array: its
// device address is looked up once per device (each GPU holds its own copy at its own address) and only that
// pointer is ever handed to a kernel.  The block reaches the array through the ARGUMENTS of small fill kernels
// (kFillChunk doubles per launch): no host buffer has to outlive the call, nothing is allocated or synchronised, and
// every launch is capturable.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU race.
constexpr int kGibbsScratchDoubles = 20480; // 160 KB: the block of d = 128 is 17 280 doubles
constexpr int kMaxDevices = 16;
static __device__ __align__(256) double g_gibbs_scratch[kGibbsScratchDoubles];

// host staging of the block while it is cut into fill arguments: static like the device array (one per device, no
// heap, no 160 KB stack frame), and like it shared by the calls on that device
static double g_gibbs_stage[kMaxDevices][kGibbsScratchDoubles];

// device address of g_gibbs_scratch on the CURRENT device, looked up once per device
static int gibbs_scratch_ptr(double **out, int *dev_out)
{
    static std::mutex mu;
    static double *ptrs[kMaxDevices] = {};
    int dev = -1;
    MCMC_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return set_error(MCMC_E_UNSUPPORTED, "no-allocation entry point: device index %d above %d", dev, kMaxDevices - 1);
    std::lock_guard<std::mutex> lock(mu);
    if (!ptrs[dev]) {
        double *ptr = nullptr;
        hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(g_gibbs_scratch));
        if (e != hipSuccess || !ptr)
            return set_error(MCMC_E_HIP, "hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
        ptrs[dev] = ptr;
    }
    *out = ptrs[dev];
    *dev_out = dev;
    return MCMC_OK;
}

int gibbs_noalloc_max_dim() // largest d whose parameter block fits the scratch array
{
    int d = kGibbsLaneMaxDim;
    while (d < kMaxDimGibbs && gibbs_block_doubles(d + 1) <= (size_t)kGibbsScratchDoubles) ++d;
    return d;
}

int launch_gibbs_noalloc(const GibbsArgs &a, const GibbsParams &p, bool tail_safe, hipStream_t s)
{
    const int d = a.d;
    if (d < 1 || d > kMaxDimGibbs) return set_error(MCMC_E_UNSUPPORTED, "gibbs: d=%d outside 1..%d", d, kMaxDimGibbs);
    const size_t total = gibbs_block_doubles(d);
    if (total > (size_t)kGibbsScratchDoubles)
        return set_error(MCMC_E_UNSUPPORTED,
                         "mcmc_tmvn_gibbs_batch_noalloc: d=%d needs a parameter block of %zu doubles, the static scratch holds %d "
                         "(d <= %d); use mcmc_tmvn_gibbs_batch",
                         d, total, kGibbsScratchDoubles, gibbs_noalloc_max_dim());
    double *ptr = nullptr;
    int dev = 0;
    int rc = gibbs_scratch_ptr(&ptr, &dev);
    if (rc != MCMC_OK) return rc;
    // One host thread at a time per device from here to the sampling kernel's launch: the staging array is shared, and the
    // fill launches of two calls must not interleave on their way into the device array.  (This orders the HOST side
    // only; two calls on different streams can still overlap on the device -- the header documents that race.)
    static std::mutex stage_mu[kMaxDevices];
    std::lock_guard<std::mutex> lock(stage_mu[dev]);
    // the whole block of this call is written (padding included), so nothing of the previous call's contents is read
    double *const blk = g_gibbs_stage[dev];
    build_gibbs_block(p, blk);
    for (size_t off = 0; off < total; off += kFillChunk) {
        const int cnt = (int)(total - off < (size_t)kFillChunk ? total - off : (size_t)kFillChunk);
        FillChunk c;
        for (int t = 0; t < cnt; ++t) c.v[t] = blk[off + t];
        for (int t = cnt; t < kFillChunk; ++t) c.v[t] = 0.0;
        hipLaunchKernelGGL(scratch_fill_kernel, dim3(1), dim3(kFillChunk), 0, s, ptr + off, cnt, c);
        hipError_t el = hipGetLastError();
        if (el != hipSuccess) return set_error(MCMC_E_HIP, "scratch_fill_kernel launch failed: %s", hipGetErrorString(el));
    }
    return launch_gibbs_on_block(a, tail_safe, ptr, s);
}

int launch_gibbs_streams(uint64_t seed, int64_t chain_offset, int64_t n_chains, int32_t n_sweeps, int32_t d,
                         double *d_U, hipStream_t s)
{
    const int block = 256;
    const int64_t grid = (n_chains + block - 1) / block;
    hipLaunchKernelGGL(gibbs_streams_kernel, dim3((unsigned)grid), dim3(block), 0, s, seed, chain_offset, n_chains,
                       n_sweeps, d, d_U);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

int launch_rtnorm(int64_t n, double mu, double sigma, double a, double b, const double *d_F, uint64_t seed,
                  int64_t offset, bool tail_safe, double *d_out, hipStream_t s)
{
    if (n == 0) return MCMC_OK;
    const int block = 256;
    const int64_t grid = (n + block - 1) / block;
    if (tail_safe)
        hipLaunchKernelGGL((rtnorm_kernel<true>), dim3((unsigned)grid), dim3(block), 0, s, n, mu, sigma, a, b, d_F, seed,
                           offset, d_out);
    else
        hipLaunchKernelGGL((rtnorm_kernel<false>), dim3((unsigned)grid), dim3(block), 0, s, n, mu, sigma, a, b, d_F, seed,
                           offset, d_out);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

} // namespace mcmc
