"""CPU suite: the C-ABI library loads, exports every symbol include/mcmc_amd.h declares, its
structs match the ctypes mirror, and the host-side logic (validation, index helpers, the shared
Gibbs precompute) works without a GPU.  No compute entry point is exercised here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mcmc_amd.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mcmc_[A-Za-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from mcmc_amd import _lib

    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} is declared in include/mcmc_amd.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes SIGNATURES and the header disagree"
    assert _lib.lib.mcmc_amd_abi_version() == 2
    assert _lib.lib.mcmc_amd_max_lane_dim() >= 10 and _lib.lib.mcmc_amd_max_dim() >= 128


def test_header_compiles_as_c_and_struct_layout_matches_ctypes(tmp_path):
    from mcmc_amd import _lib

    prog = tmp_path / "layout.c"
    prog.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "mcmc_amd.h"\n'
        "int main(void){\n"
        'printf("%zu %zu %zu %zu %zu\\n", sizeof(mcmc_target), sizeof(mcmc_rwm_config), sizeof(mcmc_rwm_io),'
        " sizeof(mcmc_gibbs_config), sizeof(mcmc_gibbs_io));\n"
        'printf("%zu %zu %zu %zu\\n", offsetof(mcmc_rwm_config, seed), offsetof(mcmc_rwm_io, d_status),'
        " offsetof(mcmc_gibbs_config, seed), offsetof(mcmc_gibbs_io, d_status));\n"
        "return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    sizes = [C.sizeof(_lib.McmcTarget), C.sizeof(_lib.RwmConfig), C.sizeof(_lib.RwmIO), C.sizeof(_lib.GibbsConfig),
             C.sizeof(_lib.GibbsIO)]
    offs = [_lib.RwmConfig.seed.offset, _lib.RwmIO.d_status.offset, _lib.GibbsConfig.seed.offset,
            _lib.GibbsIO.d_status.offset]
    assert [int(x) for x in out] == sizes + offs


def test_index_helpers_follow_the_reference(oracle):
    """sub1 / sub2 / negSubCol / negSubRow (tmvrnormGibbs.cpp:12-37) are host copies in the library."""
    import mcmc_amd

    x = np.random.default_rng(0).standard_normal((5, 5))
    for i in range(5):
        np.testing.assert_array_equal(mcmc_amd.sub1(x, i), np.delete(np.delete(x, i, 0), i, 1))
        np.testing.assert_array_equal(mcmc_amd.sub1(x, i), oracle.sub1(x, i))
        np.testing.assert_array_equal(mcmc_amd.sub2(x, i, (i + 2) % 5), oracle.sub2(x, i, (i + 2) % 5))
        np.testing.assert_array_equal(mcmc_amd.negSubCol(x[:, 0], i), np.delete(x[:, 0], i))
        np.testing.assert_array_equal(mcmc_amd.negSubRow(x[0], i), np.delete(x[0], i))
    assert mcmc_amd.sub2(x, 1, 1).shape == (1, 4)  # a 1 x (d-1) matrix, as arma::mat x.row(a)
    with pytest.raises(mcmc_amd.McmcError):
        mcmc_amd.sub1(x, 5)
    with pytest.raises(mcmc_amd.McmcError):
        mcmc_amd.negSubCol(x[0], -1)


def test_gibbs_precompute_from_precision_matches_reference_partitioning(oracle):
    import mcmc_amd

    rng = np.random.default_rng(5)
    for d in (1, 2, 5, 10, 33, 128):
        A = rng.standard_normal((d, d))
        sigma = A @ A.T / d + np.eye(d)
        P, sd = mcmc_amd.tmvn_precompute(sigma)
        st, Po, sdo = oracle.tmvn_precompute(sigma)
        assert st == 0 and P.shape == (d, d - 1)
        np.testing.assert_allclose(P, Po, rtol=0, atol=1e-13)
        np.testing.assert_allclose(sd, sdo, rtol=1e-13)
    with pytest.raises(mcmc_amd.McmcError) as e:
        mcmc_amd.tmvn_precompute(np.array([[1.0, 2.0], [2.0, 1.0]]))
    assert e.value.code == -4


def test_validation_happens_before_any_gpu_work():
    """Bad arguments are refused on the host (the reference has no validation)."""
    from mcmc_amd import _lib
    from mcmc_amd._lib import lib

    d = 2
    init = (C.c_double * d)(0.0, 0.0)
    cov = (C.c_double * 4)(1.0, 0.5, 0.2, 1.0)  # asymmetric
    prm = (C.c_double * 6)(0, 0, 1, 0, 0, 1)
    tgt = _lib.McmcTarget(_lib.MCMC_TARGET_GAUSSIAN_PREC, 6, prm)
    cfg = _lib.RwmConfig(d=d, n_iterations=10, n_chains=4, thin=1, rng_mode=0)
    io = _lib.RwmIO()
    io.h_init_theta = init
    io.h_covmat = cov
    assert lib.mcmc_adapt_rwm_batch(C.byref(tgt), C.byref(cfg), C.byref(io), None) == _lib.MCMC_E_BAD_ARG
    assert b"symmetric" in lib.mcmc_amd_last_error()
    cov2 = (C.c_double * 4)(1.0, 2.0, 2.0, 1.0)  # not positive definite
    io.h_covmat = cov2
    assert lib.mcmc_adapt_rwm_batch(C.byref(tgt), C.byref(cfg), C.byref(io), None) == _lib.MCMC_E_NOT_PD
    cfg.acceptance_window = 65537
    assert lib.mcmc_adapt_rwm_batch(C.byref(tgt), C.byref(cfg), C.byref(io), None) == _lib.MCMC_E_UNSUPPORTED
    cfg.acceptance_window = 0
    cfg.thin = 0
    assert lib.mcmc_adapt_rwm_batch(C.byref(tgt), C.byref(cfg), C.byref(io), None) == _lib.MCMC_E_BAD_ARG
    assert lib.mcmc_adapt_rwm_batch(None, C.byref(cfg), C.byref(io), None) == _lib.MCMC_E_BAD_ARG
    gcfg = _lib.GibbsConfig(d=2, n_sweeps=0, n_chains=1, thin=1)
    assert lib.mcmc_tmvn_gibbs_batch(C.byref(gcfg), C.byref(_lib.GibbsIO()), None) == _lib.MCMC_E_BAD_ARG


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under mcmc_amd/ may reference it."""
    pkg = os.path.join(ROOT, "mcmc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                for pat in ("liboracle", "mcmc_oracle", "from oracle", "import oracle", "oracle/", "orc_"):
                    assert pat not in txt, f"{f} references the oracle ({pat})"


def test_missing_library_fails_loudly(tmp_path):
    """Importing the package without libmcmc_amd.so must raise, not fall back to the CPU."""
    code = (
        "import importlib.util, sys, os\n"
        f"src = os.path.join({ROOT!r}, 'mcmc_amd', '_lib.py')\n"
        f"dst = os.path.join({str(tmp_path)!r}, '_lib.py')\n"
        "open(dst, 'w').write(open(src).read())\n"
        "spec = importlib.util.spec_from_file_location('_lib_copy', dst)\n"
        "m = importlib.util.module_from_spec(spec)\n"
        "try:\n    spec.loader.exec_module(m)\nexcept ImportError as e:\n    print('IMPORTERROR', e); sys.exit(0)\n"
        "sys.exit(1)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and "IMPORTERROR" in r.stdout and "no CPU fallback" in r.stdout


def test_melt_trace_matches_the_reference_layout():
    """melt_trace (MCMCutilty.R:19-35): parameters, then `chain` (1-based factor) and `iteration`, stacked chain by chain."""
    import mcmc_amd

    rng = np.random.default_rng(0)
    tr = rng.standard_normal((5, 3, 4))  # [n_kept, d, n_chains]
    chains = mcmc_amd.chain_list(tr, ["a", "b", "c"], acceptance_rate=np.array([0.1, 0.2, 0.3, 0.4]))
    assert len(chains) == 4 and chains[2]["trace"].shape == (5, 3) and chains[3]["acceptance_rate"] == 0.4
    df = mcmc_amd.melt_trace(chains)
    assert list(df.columns) == ["a", "b", "c", "chain", "iteration"] and len(df) == 20
    np.testing.assert_array_equal(df["iteration"].to_numpy(), np.tile(np.arange(1, 6), 4))
    np.testing.assert_array_equal(np.asarray(df["chain"]), np.repeat(np.arange(1, 5), 5))
    np.testing.assert_array_equal(df[["a", "b", "c"]].to_numpy()[5:10], tr[:, :, 1])
    assert str(df["chain"].dtype) == "category"
    # default column names are R's V1..Vd; a subset of chains can be selected before the host copy
    sub = mcmc_amd.chain_list(tr, chains=[3, 0])
    assert sub[0]["par_names"] == ["V1", "V2", "V3"]
    np.testing.assert_array_equal(sub[0]["trace"], tr[:, :, 3])
    with pytest.raises(ValueError):
        mcmc_amd.melt_trace([])


def test_user_target_compiles_without_a_gpu_and_reports_diagnostics():
    """mcmc_target_compile: hipRTC cross-compiles for gfx950 with no device."""
    import mcmc_amd

    src = """
__device__ double mcmc_user_target(const double (&x)[MCMC_USER_D], const double *params)
{
    double s = 0.0;
    for (int k = 0; k < MCMC_USER_D; ++k) s += params[0] * x[k] * x[k];
    return -0.5 * s;
}
"""
    t = mcmc_amd.Target.compile(src, 3, [2.0])
    assert t.kind == 3 and t._user.value
    c = t._c()
    assert c.kind == 3 and c.n_params == 1 and c.user == t._user.value
    with pytest.raises(mcmc_amd.McmcError, match="undeclared identifier 'y'"):
        mcmc_amd.Target.compile("__device__ double mcmc_user_target(const double (&x)[MCMC_USER_D], const double *p) { return y; }", 2)
    with pytest.raises(mcmc_amd.McmcError, match="mcmc_user_target_part"):
        mcmc_amd.Target.compile(src, 13, [2.0])  # d > 12 runs per workgroup and needs the per-part form
    with pytest.raises(mcmc_amd.McmcError, match="outside 1..128"):
        mcmc_amd.Target.compile(src, 129, [2.0])
    with pytest.raises(ValueError):
        t.with_params([1.0, 2.0])


def test_hand_written_dpp_blocks_keep_the_valu_write_hazard_distance():
    """ADVICE r2 (rwm_row.hip): the v_fmac_f64_dpp asm blocks rely on an s_nop placed where their operand is created; the
    compiler does not see DPP reads inside inline asm.  tools/check_dpp_hazard.py disassembles the built gfx950 code and
    fails if any DPP instruction reads a VGPR that a VALU instruction wrote within the two preceding wait states."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("check_dpp_hazard", os.path.join(ROOT, "tools", "check_dpp_hazard.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    # the checker itself: a VALU write followed at once by a DPP read of that register is a violation, two wait states cure it
    write = ("v_mul_f64", ["v[20:21]", "v[2:3]", "v[4:5]"])
    dpp = ("v_fmac_f64_dpp", ["v[22:23]", "v[20:21]", "v[6:7] row_newbcast:0 row_mask:0xf bank_mask:0xf"])
    other = ("v_add_f64", ["v[30:31]", "v[2:3]", "v[4:5]"])
    assert len(chk.violations([write, dpp])) == 1
    assert len(chk.violations([write, other, dpp])) == 1
    assert chk.violations([write, other, other, dpp]) == []
    assert chk.violations([write, ("s_nop", ["1"]), dpp]) == []
    assert len(chk.violations([write, ("s_nop", ["0"]), dpp])) == 1
    assert chk.violations([("ds_read_b64", ["v[20:21]", "v9"]), dpp]) == []  # not a VALU write: waitcnt's business
    if not os.path.exists(chk.OBJDUMP):
        pytest.skip("no llvm-objdump")
    for name in ("rwm_row.o", "rwm_block.o"):
        obj = os.path.join(ROOT, "mcmc_amd", "csrc", name)
        if not os.path.exists(obj):
            pytest.skip(f"{name} not built in-tree")
        insts = chk.disassemble(obj)
        n_dpp = sum(1 for mn, _ in insts if mn.endswith("_dpp"))
        assert n_dpp > 100, (name, n_dpp)
        assert chk.violations(insts) == [], name
    # and the fences are what keeps it so: without the s_nops the same code WOULD violate the rule
    insts = [i for i in chk.disassemble(os.path.join(ROOT, "mcmc_amd", "csrc", "rwm_row.o")) if i[0] != "s_nop"]
    assert len(chk.violations(insts)) > 0
