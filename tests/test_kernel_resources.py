"""What the compiler made of the hot kernels (no GPU needed: hipcc cross-compiles gfx950): registers, spills, scratch, LDS and
the occupancy they admit, from `make -C tyrant_amd/csrc asm` (-Rpass-analysis=kernel-resource-usage).

Round 4's ISA audit found k_shade holding 113 scalar values in vector lanes and running at four blocks per CU
because a by-value kernel argument is loaded in the kernel's first block and kept for its whole life; reading it where it lies
(device_common.hpp kernarg_view) brought that to no spill at all and five blocks per CU, worth 15 % of the kernel.  A change
that quietly brings the spills back, or one LDS granule too many (the CU then places four blocks where the occupancy query
still answers five), costs that again without failing any parity test: this file is the guard."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tyrant_amd", "csrc")
LDS_PER_CU = 163840  # MI355X: 160 KB per CU ...
LDS_GRANULE = 1280   # ... handed out in 1280-byte granules (hip/traverse_flat.hip k_trace_flat: 31,748 B x 5 fit, 32,004 B x 5 do not)


def _resources(unit):
    out = {}
    cur = None
    for line in open(os.path.join(CSRC, "build", unit + ".resources.txt")):
        m = re.search(r"remark:\s+(.*?)\s+\[-Rpass-analysis", line)
        if not m:
            continue
        text = m.group(1)
        if text.startswith("Function Name:"):
            cur = out.setdefault(text.split(":", 1)[1].strip(), {})
        elif cur is not None and ":" in text:
            k, v = text.rsplit(":", 1)
            cur[k.strip()] = v.strip()
    return out


@pytest.fixture(scope="module")
def resources():
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    subprocess.run(["make", "-s", "-C", CSRC, "asm"], check=True, capture_output=True, timeout=900)
    r = {}
    for unit in ("frame", "shade", "traverse_flat"):
        r.update(_resources(unit))
    return r


def _kernel(resources, key):
    names = [n for n in resources if key in n]
    assert len(names) == 1, (key, names)
    return {k: (int(v) if v.lstrip("-").isdigit() else v) for k, v in resources[names[0]].items()}


def _blocks_that_fit(lds_bytes):
    per_block = -(-lds_bytes // LDS_GRANULE) * LDS_GRANULE
    return LDS_PER_CU // per_block


@pytest.mark.parametrize("key", ["k_shadeILb0EEE", "k_shadeILb1EEE"])
def test_k_shade_keeps_nothing_in_spills_and_runs_five_blocks_per_cu(resources, key):
    k = _kernel(resources, key)
    assert k["SGPRs Spill"] == 0 and k["VGPRs Spill"] == 0 and k["ScratchSize [bytes/lane]"] == 0, k  # (a scratch reload waits on vmcnt behind the pixel atomics this kernel leaves in flight)
    assert k["Occupancy [waves/SIMD]"] >= 5 and k["VGPRs"] + k["AGPRs"] <= 96, k  # five blocks of four waves per CU = five waves per SIMD
    assert _blocks_that_fit(k["LDS Size [bytes/block]"]) >= 5, k


def test_k_trace_flat_fits_five_blocks_per_cu_without_vector_spills(resources):
    k = _kernel(resources, "k_trace_flatILi12ELj256E")
    assert k["VGPRs Spill"] == 0 and k["SGPRs Spill"] <= 4, k  # (24 scalar spills before the refill / flush / end read the argument through the view)
    assert k["Occupancy [waves/SIMD]"] >= 5 and k["VGPRs"] + k["AGPRs"] <= 96, k
    assert _blocks_that_fit(k["LDS Size [bytes/block]"]) >= 5, k  # one granule more and the persistent grid's fifth blocks run after the others (+30 % per render, round 2)


def test_k_trace_flat_in_768_thread_blocks_is_six_waves_per_simd(resources):
    # two 12-wave blocks per CU share two copies of the staged nodes instead of five: 2 x (73,728 + 7,168 + 4) B is ALL of the CU's LDS
    k = _kernel(resources, "k_trace_flatILi12ELj768E")
    assert k["VGPRs Spill"] == 0 and k["SGPRs Spill"] <= 4, k
    assert k["Occupancy [waves/SIMD]"] >= 6 and k["VGPRs"] + k["AGPRs"] <= 80, k
    assert _blocks_that_fit(k["LDS Size [bytes/block]"]) >= 2, k


def test_k_primary_has_no_spills(resources):
    k = _kernel(resources, "k_primaryENS")
    assert k["SGPRs Spill"] == 0 and k["VGPRs Spill"] == 0 and k["ScratchSize [bytes/lane]"] == 0, k
