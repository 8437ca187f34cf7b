"""The C-ABI library loads and exports every symbol include/*.h declares; no compute without a GPU."""
import ctypes as C
import os
import re
import sys

import pytest

import delphy_amd as d

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for hdr in ("emat_backend.h", "emat_host.h", "emat_dphy.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(?:emat_status|const char\*|void|int32_t|emat_backend\*|emat_run\*)\s+(emat_\w+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(d.library_path())
    syms = declared_symbols()
    assert len(syms) >= 45
    for s in syms:
        assert hasattr(lib, s), "missing export: " + s


def test_no_cpu_fallback():
    import torch
    if torch.cuda.device_count() == 0:
        with pytest.raises(d.EmatError, match="NO_DEVICE"):
            d.EmatBackend(1000)                  # a real handle needs a HIP device
    b = d.EmatBackend(1000, device=-1)           # host-only handle (also on a GPU box): staging works, launches do not
    with pytest.raises(d.EmatError):
        b.run_moves_per_part(1)
    with pytest.raises(d.EmatError, match="NO_DEVICE"):
        b.global_stats(1)                        # device-side reductions have no host stand-in either
    with pytest.raises(d.EmatError):
        b.recalc_derived()
    with pytest.raises(d.EmatError, match="NO_DEVICE"):
        b.num_muts_l()
    with pytest.raises(d.EmatError, match="NO_DEVICE"):
        b.scalable_coalescent_log_prior(0.0, 1.0)
    with pytest.raises(d.EmatError, match="NO_DEVICE"):
        b.debug_gamma(0, [1.0], [1.0])
    b.close()


def test_product_never_references_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "delphy_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"oracle_ffi|orc_capi|libemat_oracle|#include\s+\"[^\"]*orc_", txt):
                    bad.append(f)
    assert not bad, bad


def test_the_hbm_resident_tree_has_no_host_fallback():
    """emat_tree_* on a handle without a device: loud EMAT_ERR_NO_DEVICE, never a CPU path."""
    import numpy as np
    import delphy_amd as d
    from delphy_amd.scenarios import make_scenario
    sc = make_scenario("C1", num_tips=20, num_sites=300)
    b = d.EmatBackend(sc.num_sites, device=-1)
    run = d.EmatRun(b, sc.tree, sc.ref, 1)
    run.set_num_parts(2); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop)
    run.set_device_tree(True)
    with pytest.raises(d.EmatError, match="EMAT_ERR_NO_DEVICE"):
        run.repartition()
    run.close(); b.close()


def test_options_are_set_per_handle_and_the_library_reads_no_tuning_from_the_environment():
    """emat_set_option replaces the EMAT_* environment switches the library used to read at emat_backend_create:
    known names are taken, unknown ones refused, and the only getenv left in the product's C++ is EMAT_VERBOSE."""
    b = d.EmatBackend(100, device=-1)
    for k, v in (("lds_classes", "60,90"), ("chunks", 3), ("ticket_release", "full"), ("tree_host_coalescent", 1), ("slack", 1.5)):
        b.set_option(k, v)
    with pytest.raises(d.EmatError, match="unknown option"):
        b.set_option("no_such_option", 1)
    b.close()
    left = set()
    for f in os.listdir(os.path.join(ROOT, "delphy_amd", "csrc")):
        if f.endswith((".hpp", ".hip", ".cpp")):
            left |= set(re.findall(r'getenv\("(\w+)"\)', open(os.path.join(ROOT, "delphy_amd", "csrc", f)).read()))
    assert left == {"EMAT_VERBOSE"}, left


def test_a_missing_rccl_is_reported_not_crashed_on():
    """Rccl::load with a library that cannot be opened (the message was built from two dlerror() calls, the second
    of which returns NULL): a status and a text, and the usual names work again afterwards."""
    lib = d.load_library()
    buf = C.create_string_buffer(512)
    assert lib.emat_multi_set_rccl_library(b"/nonexistent/librccl_not_here.so") == 0
    try:
        assert lib.emat_multi_debug_rccl_load(buf, 512) != 0
        assert b"could not be loaded" in buf.value and b"librccl_not_here" in buf.value, buf.value
    finally:
        lib.emat_multi_set_rccl_library(None)
    assert lib.emat_multi_debug_rccl_load(buf, 512) == 0, buf.value      # the image ships RCCL: the default names load


def test_state_frequencies_of_the_reference_sequence_per_site_partition():
    """emat_part_get_state_frequencies = Subrun::state_frequencies_of_ref_sequence_per_partition (subrun.h:45, read at run.cpp:354-355;
    calc_state_frequencies_per_partition_of, phylo_tree_calc.cpp:95-106): counts[site partition][state] of the reference sequence the
    parts are written against -- with one and with two site partitions, on a host-only handle (it is a read-back, no launch)."""
    import numpy as np
    from delphy_amd.scenarios import make_scenario
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import configure, split_parts
    sc = make_scenario("C1", num_tips=30, num_sites=900)
    parts, incl, seeds, root_part, ref = split_parts(sc, 3, 5)
    b = d.EmatBackend(sc.num_sites, device=-1)
    configure(b, sc, ref, parts, incl, seeds, root_part)
    for p in range(len(parts)):
        got = b.part_state_frequencies(p)
        assert got.shape == (1, 4) and np.array_equal(got[0], np.bincount(ref, minlength=4)) and got.sum() == sc.num_sites
    b.close()
    # two site partitions (the mpox shape, run.cpp:400-435)
    pfs = (np.arange(sc.num_sites) % 3 == 0).astype(np.uint8)
    mu = np.array([1e-3, 2e-3]); pi = np.array([[0.31, 0.19, 0.21, 0.29], [0.25, 0.25, 0.25, 0.25]])
    q = np.zeros((2, 4, 4))
    for k in range(2):
        for a in range(4):
            for c in range(4):
                if a != c: q[k, a, c] = pi[k, c]
            q[k, a, a] = -q[k, a].sum()
    b = d.EmatBackend(sc.num_sites, device=-1)
    configure(b, sc, ref, parts, incl, seeds, root_part, evo=(mu, pi, q, pfs))
    got = b.part_state_frequencies(root_part)
    want = np.stack([np.bincount(ref[pfs == k], minlength=4) for k in range(2)])
    assert np.array_equal(got, want)
    b.close()
