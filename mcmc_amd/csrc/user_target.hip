// User targets: a log-density supplied as HIP source, compiled at run time with hipRTC together with
// the one-chain-per-lane Metropolis kernels, so that the function is inlined into the sampling loop.
//
// This is the compiled-code replacement for an arbitrary R closure `target`
// (adaptMCMC_source.cpp:75,97,158; :230,246,277), which a GPU cannot call.
// The run-time translation unit is
//     #define MCMC_USER_D <d> / MCMC_USER_NP <n_params>
//     #include "rwm_lane.hip"   (device part only: __HIPCC_RTC__ hides the host launchers)
//     #include "rwm_pair.hip"   (d <= 10)
//     <the user's source, defining mcmc_user_target>
// and the kernels are the same templates the library instantiates for its built-in targets, with
// KIND = MCMC_TARGET_USER.  The host side (parameter block, grid, LDS size) is shared with them
// (launch_rwm_lane_with / launch_rwm_pair_with).
#include <dlfcn.h>
#include <hip/hiprtc.h>

#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>

#include "common.h"
#include "targets.h"

namespace mcmc {

namespace {

constexpr uint32_t kUserMagic = 0x55544721u;

struct UserModule {
    hipModule_t mod = nullptr;
    hipFunction_t lane[2] = {nullptr, nullptr}; // FULL = false / true
    hipFunction_t pair[3] = {nullptr, nullptr, nullptr}; // no moments / MOMENTS / MOMENTS + STATE
    hipFunction_t block[2] = {nullptr, nullptr}; // d > 12: FULL = false / true
    hipFunction_t eval = nullptr;
};

struct UserTarget {
    uint32_t magic = kUserMagic;
    int d = 0, np = 0;
    bool has_pair = false, is_block = false; // is_block: d > 12, compiled into the per-workgroup kernel
    std::vector<char> code;
    std::string lane_name[2], pair_name[3], block_name[2], eval_name;
    std::mutex mu;
    std::map<int, UserModule> per_device; // loaded at first use on each device
};

UserTarget *as_user(const void *h)
{
    UserTarget *u = (UserTarget *)h;
    return (u && u->magic == kUserMagic) ? u : nullptr;
}

std::string env_or(const char *name, const std::string &fallback)
{
    const char *v = std::getenv(name);
    return (v && *v) ? std::string(v) : fallback;
}

// directory of this shared library (mcmc_amd/ in the source tree)
std::string library_dir()
{
    Dl_info info;
    if (dladdr((const void *)&as_user, &info) && info.dli_fname) {
        std::string f(info.dli_fname);
        const size_t slash = f.rfind('/');
        return slash == std::string::npos ? std::string(".") : f.substr(0, slash);
    }
    return ".";
}

int get_module(UserTarget *u, UserModule **out)
{
    int dev = 0;
    MCMC_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(u->mu);
    UserModule &m = u->per_device[dev];
    if (!m.mod) {
        MCMC_HIP_CHECK(hipModuleLoadData(&m.mod, u->code.data()));
        for (int v = 0; v < 2 && u->is_block; ++v) MCMC_HIP_CHECK(hipModuleGetFunction(&m.block[v], m.mod, u->block_name[v].c_str()));
        for (int v = 0; v < 2 && !u->is_block; ++v) MCMC_HIP_CHECK(hipModuleGetFunction(&m.lane[v], m.mod, u->lane_name[v].c_str()));
        for (int v = 0; v < 3 && u->has_pair; ++v)
            MCMC_HIP_CHECK(hipModuleGetFunction(&m.pair[v], m.mod, u->pair_name[v].c_str()));
        MCMC_HIP_CHECK(hipModuleGetFunction(&m.eval, m.mod, u->eval_name.c_str()));
    }
    *out = &m;
    return MCMC_OK;
}

hipError_t module_launch(hipFunction_t f, unsigned grid, unsigned block, size_t ldsb, hipStream_t s, const RwmArgs &a,
                         const double *d_prm)
{
    RwmArgs aa = a;
    void *params[] = {&aa, &d_prm};
    return hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, (unsigned)ldsb, s, params, nullptr);
}

} // namespace

int user_target_param_count(const void *handle, int d)
{
    const UserTarget *u = as_user(handle);
    if (!u) return set_error(MCMC_E_BAD_ARG, "MCMC_TARGET_USER needs the handle mcmc_target_compile returned");
    if (u->d != d) return set_error(MCMC_E_BAD_ARG, "user target was compiled for d=%d, called with d=%d", u->d, d);
    return u->np;
}

int launch_rwm_block_user(const RwmArgs &a, const RwmParams &p, hipStream_t s)
{
    UserTarget *u = as_user(p.user);
    if (!u || u->d != a.d) return set_error(MCMC_E_BAD_ARG, "user target handle missing or compiled for another d");
    if (!u->is_block)
        return set_error(MCMC_E_UNSUPPORTED, "this user target was compiled for the one-chain-per-lane kernels (d <= %d)", kMaxLaneDimRwm);
    UserModule *m = nullptr;
    const int rc = get_module(u, &m);
    if (rc != MCMC_OK) return rc;
    return launch_rwm_block_with(a, p, s, true,
                                 [m](int full, unsigned grid, unsigned block, size_t ldsb, hipStream_t st, const RwmArgs &aa,
                                     const double *d_prm) { return module_launch(m->block[full], grid, block, ldsb, st, aa, d_prm); });
}

int launch_rwm_lane_user(const RwmArgs &a, const RwmParams &p, hipStream_t s)
{
    UserTarget *u = as_user(p.user);
    if (!u || u->d != a.d) return set_error(MCMC_E_BAD_ARG, "user target handle missing or compiled for another d");
    if (u->is_block)
        return set_error(MCMC_E_UNSUPPORTED, "this user target was compiled for the per-workgroup kernel (d > %d)", kMaxLaneDimRwm);
    UserModule *m = nullptr;
    const int rc = get_module(u, &m);
    if (rc != MCMC_OK) return rc;
    return launch_rwm_lane_with(a, p, s, lane_layout(a.d, u->np),
                                [m](int full, unsigned grid, unsigned block, size_t ldsb, hipStream_t st, const RwmArgs &aa,
                                    const double *d_prm) { return module_launch(m->lane[full], grid, block, ldsb, st, aa, d_prm); });
}

int launch_rwm_pair_user(const RwmArgs &a, const RwmParams &p, hipStream_t s)
{
    UserTarget *u = as_user(p.user);
    if (!u || u->d != a.d) return set_error(MCMC_E_BAD_ARG, "user target handle missing or compiled for another d");
    if (!u->has_pair) return set_error(MCMC_E_UNSUPPORTED, "paired-wave kernel: d=%d outside 1..%d", a.d, kMaxPairDimRwm);
    UserModule *m = nullptr;
    const int rc = get_module(u, &m);
    if (rc != MCMC_OK) return rc;
    return launch_rwm_pair_with(a, p, s, u->np, p.tparams,
                                [m](int moments, unsigned grid, unsigned block, size_t ldsb, hipStream_t st, const RwmArgs &aa,
                                    const double *d_prm) { return module_launch(m->pair[moments], grid, block, ldsb, st, aa, d_prm); });
}

// built-in targets: the eval kernel is compiled into the library
template <int D, int KIND>
static hipError_t eval_builtin(int64_t n, const double *d_x, const double *d_prm, double *d_out, hipStream_t s)
{
    hipLaunchKernelGGL((target_eval_kernel<D, KIND>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, d_x, d_prm, d_out);
    return hipGetLastError();
}

} // namespace mcmc

using namespace mcmc;

extern "C" int mcmc_target_compile(const char *source, int32_t d, int32_t n_params, void **handle, char *log,
                                   int64_t log_capacity)
{
    if (log && log_capacity > 0) log[0] = 0;
    if (!source || !handle) return set_error(MCMC_E_BAD_ARG, "mcmc_target_compile: NULL argument");
    *handle = nullptr;
    if (d < 1 || d > kMaxDimRwm) return set_error(MCMC_E_UNSUPPORTED, "user targets: d=%d outside 1..%d", d, kMaxDimRwm);
    const bool is_block = d > kMaxLaneDimRwm;
    // one chain per lane: the parameters sit in LDS beside the shared block; per workgroup: in HBM / L2 (room for data)
    const int max_params = is_block ? (1 << 22) : 4096;
    if (n_params < 0 || n_params > max_params) return set_error(MCMC_E_BAD_ARG, "n_params=%d outside 0..%d", n_params, max_params);

    if (is_block && !std::strstr(source, "mcmc_user_target_part"))
        return set_error(MCMC_E_BAD_ARG, "d=%d > %d runs one chain per workgroup: the source has to define "
                         "mcmc_user_target_part(const double *x, int d, const double *params, int part, int nparts)", d, kMaxLaneDimRwm);
    if (!is_block && !std::strstr(source, "mcmc_user_target"))
        return set_error(MCMC_E_BAD_ARG, "the source has to define mcmc_user_target(const double (&x)[MCMC_USER_D], const double *params)");
    const std::string libdir = library_dir();
    const std::string srcdir = env_or("MCMC_AMD_SRC_DIR", libdir + "/csrc");
    const std::string incdir = env_or("MCMC_AMD_INCLUDE_DIR", libdir + "/../include");
    const std::string rocm = env_or("ROCM_PATH", "/opt/rocm");
    const bool has_pair = d <= kMaxPairDimRwm;
    const int nt = (d + 15) / 16;

    // the run-time compiler has no <stdint.h>; it keeps its own fixed-width types in a namespace
    std::string tu =
        "typedef signed char int8_t;\ntypedef unsigned char uint8_t;\ntypedef signed int int32_t;\ntypedef unsigned int uint32_t;\n"
        "typedef signed long int64_t;\ntypedef unsigned long uint64_t;\n";
    tu += "#define MCMC_USER_NP " + std::to_string(n_params) + "\n";
    if (is_block) {
        tu += "#define MCMC_USER_BLOCK 1\n#define MCMC_USER_D " + std::to_string(d) + "\n#include \"rwm_block.hip\"\n";
    } else {
        tu += "#define MCMC_USER_D " + std::to_string(d) + "\n#include \"rwm_lane.hip\"\n";
        if (has_pair) tu += "#include \"rwm_pair.hip\"\n";
    }
    tu += "#line 1 \"mcmc_user_target\"\n";
    tu += source;
    tu += "\n";

    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, tu.c_str(), "mcmc_user_target_module.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        return set_error(MCMC_E_HIP, "hiprtcCreateProgram failed");
    auto up = std::make_unique<UserTarget>();
    up->d = d;
    up->np = n_params;
    up->has_pair = has_pair;
    up->is_block = is_block;
    const std::string ds = std::to_string(d), ks = std::to_string(MCMC_TARGET_USER);
    std::string expr_lane[2], expr_pair[3], expr_block[2];
    const std::string expr_eval = is_block ? std::string("mcmc::block_target_eval_kernel") : "mcmc::target_eval_kernel<" + ds + ", " + ks + ">";
    for (int v = 0; v < 2; ++v) {
        expr_lane[v] = "mcmc::rwm_lane_kernel<" + ds + ", " + ks + ", " + (v ? "true" : "false") + ">";
        expr_block[v] = "mcmc::rwm_block_kernel<" + std::to_string(nt) + ", " + (v ? "true" : "false") + ", true>";
        hiprtcAddNameExpression(prog, is_block ? expr_block[v].c_str() : expr_lane[v].c_str());
    }
    for (int v = 0; v < 3 && has_pair; ++v) {
        expr_pair[v] = "mcmc::rwm_pair_kernel<" + ds + ", " + ks + ", " + (v ? "true" : "false") + ", " + (v == 2 ? "true" : "false") + ", true>"; // WIN = true: the general variant
        hiprtcAddNameExpression(prog, expr_pair[v].c_str());
    }
    hiprtcAddNameExpression(prog, expr_eval.c_str());

    const std::string i_src = "-I" + srcdir, i_inc = "-I" + incdir, i_rocm = "-I" + rocm + "/include";
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", i_src.c_str(), i_inc.c_str(), i_rocm.c_str()};
    const hiprtcResult res = hiprtcCompileProgram(prog, (int)(sizeof opts / sizeof opts[0]), opts);
    size_t log_size = 0;
    hiprtcGetProgramLogSize(prog, &log_size);
    std::string build_log(log_size, '\0');
    if (log_size) hiprtcGetProgramLog(prog, &build_log[0]);
    if (log && log_capacity > 0) {
        const size_t ncopy = std::min((size_t)log_capacity - 1, build_log.size());
        std::memcpy(log, build_log.data(), ncopy);
        log[ncopy] = 0;
    }
    if (res != HIPRTC_SUCCESS) {
        hiprtcDestroyProgram(&prog);
        std::string first = build_log.substr(0, 300);
        return set_error(MCMC_E_BAD_ARG, "user target did not compile (%s): %s", hiprtcGetErrorString(res), first.c_str());
    }
    bool ok = true;
    auto lowered = [&](const std::string &expr, std::string &out) {
        const char *nm = nullptr;
        if (hiprtcGetLoweredName(prog, expr.c_str(), &nm) != HIPRTC_SUCCESS || !nm) ok = false;
        else out = nm;
    };
    for (int v = 0; v < 2; ++v) {
        if (is_block) lowered(expr_block[v], up->block_name[v]);
        else lowered(expr_lane[v], up->lane_name[v]);
    }
    for (int v = 0; v < 3 && has_pair; ++v) lowered(expr_pair[v], up->pair_name[v]);
    lowered(expr_eval, up->eval_name);
    size_t code_size = 0;
    if (ok && hiprtcGetCodeSize(prog, &code_size) == HIPRTC_SUCCESS && code_size) {
        up->code.resize(code_size);
        ok = hiprtcGetCode(prog, up->code.data()) == HIPRTC_SUCCESS;
    } else {
        ok = false;
    }
    hiprtcDestroyProgram(&prog);
    if (!ok) return set_error(MCMC_E_HIP, "hipRTC produced no code object for the user target");
    *handle = up.release();
    return MCMC_OK;
}

extern "C" int mcmc_target_destroy(void *handle)
{
    UserTarget *u = as_user(handle);
    if (!u) return set_error(MCMC_E_BAD_ARG, "mcmc_target_destroy: not a user target handle");
    for (auto &kv : u->per_device)
        if (kv.second.mod) (void)hipModuleUnload(kv.second.mod);
    u->magic = 0;
    delete u;
    return MCMC_OK;
}

extern "C" int mcmc_target_eval_batch(const mcmc_target *target, int32_t d, int64_t n, const double *d_x, double *d_out,
                                      void *stream)
{
    if (!target || !d_x || !d_out || n < 1) return set_error(MCMC_E_BAD_ARG, "mcmc_target_eval_batch: bad argument");
    const bool wide = d > kMaxLaneDimRwm;
    if (d < 1 || d > kMaxDimRwm || (wide && target->kind != MCMC_TARGET_USER))
        return set_error(MCMC_E_UNSUPPORTED, "mcmc_target_eval_batch: built-in targets d <= %d, user targets d <= %d (d=%d)",
                         kMaxLaneDimRwm, kMaxDimRwm, d);
    hipStream_t s = (hipStream_t)stream;
    int np;
    if (target->kind == MCMC_TARGET_USER) {
        np = user_target_param_count(target->user, d);
        if (np < 0) return np;
    } else if (target->kind >= 0 && target->kind <= MCMC_TARGET_BANANA) {
        np = target_param_count(target->kind, d);
        if (target->kind != MCMC_TARGET_GAUSSIAN_PREC && d != 2) return set_error(MCMC_E_BAD_ARG, "target kind %d is 2-dimensional (d=%d)", target->kind, d);
    } else {
        return set_error(MCMC_E_BAD_ARG, "unknown target kind %d", target->kind);
    }
    if (target->n_params != np || (np > 0 && !target->h_params))
        return set_error(MCMC_E_BAD_ARG, "target kind %d with d=%d needs %d parameters, got %d", target->kind, d, np, target->n_params);
    double *d_prm = nullptr;
    MCMC_HIP_CHECK(hipMallocAsync((void **)&d_prm, sizeof(double) * (np ? np : 1), s));
    if (np) MCMC_HIP_CHECK(hipMemcpyAsync(d_prm, target->h_params, sizeof(double) * np, hipMemcpyHostToDevice, s));
    MCMC_HIP_CHECK(hipStreamSynchronize(s)); // h_params belongs to the caller
    hipError_t e1 = hipSuccess;
    int rc = MCMC_OK;
    if (target->kind == MCMC_TARGET_USER) {
        UserModule *m = nullptr;
        rc = get_module(as_user(target->user), &m);
        if (rc == MCMC_OK) {
            int64_t nn = n;
            int dd = d;
            const double *x = d_x, *prm = d_prm;
            double *out = d_out;
            if (wide) { // one workgroup per point
                void *params[] = {&nn, &dd, &x, &prm, &out};
                e1 = hipModuleLaunchKernel(m->eval, (unsigned)n, 1, 1, 256, 1, 1, (unsigned)((d + 8) * sizeof(double)), s, params, nullptr);
            } else {
                void *params[] = {&nn, &x, &prm, &out};
                e1 = hipModuleLaunchKernel(m->eval, (unsigned)((n + 255) / 256), 1, 1, 256, 1, 1, 0, s, params, nullptr);
            }
        }
    } else if (target->kind == MCMC_TARGET_GOLDSTEIN_PRICE) {
        e1 = eval_builtin<2, MCMC_TARGET_GOLDSTEIN_PRICE>(n, d_x, d_prm, d_out, s);
    } else if (target->kind == MCMC_TARGET_BANANA) {
        e1 = eval_builtin<2, MCMC_TARGET_BANANA>(n, d_x, d_prm, d_out, s);
    } else {
        switch (d) {
#define MCMC_CASE(D) \
    case D: e1 = eval_builtin<D, MCMC_TARGET_GAUSSIAN_PREC>(n, d_x, d_prm, d_out, s); break;
            MCMC_CASE(1) MCMC_CASE(2) MCMC_CASE(3) MCMC_CASE(4) MCMC_CASE(5) MCMC_CASE(6)
            MCMC_CASE(7) MCMC_CASE(8) MCMC_CASE(9) MCMC_CASE(10) MCMC_CASE(11) MCMC_CASE(12)
#undef MCMC_CASE
        }
    }
    hipError_t e2 = hipFreeAsync(d_prm, s);
    if (rc != MCMC_OK) return rc;
    if (e1 != hipSuccess) return set_error(MCMC_E_HIP, "target_eval_kernel launch failed: %s", hipGetErrorString(e1));
    if (e2 != hipSuccess) return set_error(MCMC_E_HIP, "hipFreeAsync failed: %s", hipGetErrorString(e2));
    return MCMC_OK;
}
