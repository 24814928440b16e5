"""Every kernel variant the dispatch can choose, FORCED onto the kernel parity suites.

Which instantiation runs for a launch is decided by size thresholds (wdpm_launch_fused_rows / wdpm_launch_small_rows), so an
ordinary run of the suite meets each variant only on the sizes the dispatch gives it.  Rounds 1 - 3 forced them by hand (32
tools/gpu_r03_*.sh sessions the driver's GPU test never saw).  Here each combination of the library's A/B switches runs the
stencil golden vectors, the random rasters, the adversarial operand pools, every outlet position of a window and the clamped-step
cases in a child process (the switches are read once per process), bit for bit against the oracle and the reference's vectors:

  gated, unclamped, no issue priorities      WDPM_PLAIN=0 WDPM_CLAMP=0 WDPM_PRIO=0 (and no 16-bit DEM offsets: WDPM_DEM16=0)
  the marching kernel on every size          WDPM_RELAY=0 WDPM_TRI=0, DEM codes on every launch (WDPM_DEM32=2), chunk heights from
                                             deliberately skewed per-XCD weights (WDPM_BALANCE=2)
  the relay kernel on every size             WDPM_RELAY=2, four-wave and eight-wave workgroups, stage priorities forced
  the triangle kernel on every size          WDPM_TRI=2 WDPM_RELAY=0, three and six rows per wave
  (round 5) the marching kernel on every size with every slot of the resident round filled - chunks down to two row triples - and the
  row boundaries of partner strips rounded half a triple apart    WDPM_BALANCE=2 WDPM_PAIR=2"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SUITE = ["tests/test_hip_parity.py::test_golden_stencil_vectors", "tests/test_hip_parity.py::test_random_rasters_match_oracle",
         "tests/test_hip_parity.py::test_dem_codes_on_and_off", "tests/test_hip_parity.py::test_dem_codes_as_16_bit_offsets_on_and_off",
         "tests/test_hip_parity.py::test_adversarial_operands", "tests/test_hip_parity.py::test_drain_on_codes_with_nodata_around_the_outlet",
         "tests/test_hip_parity.py::test_steady_iterations_of_small_rasters_replayed_as_hip_graphs[120-300-add]",
         "tests/test_hip_parity.py::test_steady_iterations_of_small_rasters_replayed_as_hip_graphs[120-300-drain]",
         "tests/test_hip_parity.py::test_drain_outlet_at_every_window_position", "tests/test_hip_parity.py::test_block_loop_matches_oracle",
         "tests/test_hip_parity.py::test_every_height_around_chunk_boundaries", "tests/test_hip_parity.py::test_every_width_around_strip_boundaries",
         "tests/test_hip_parity.py::test_negative_and_nan_inputs_are_handled_like_the_reference", "tests/test_hip_parity.py::test_degenerate_shapes",
         "tests/test_hip_parity.py::test_triangle_kernel_in_several_rounds", "tests/test_clamped_step.py"]

VARIANTS = {
    "gated-unclamped-no-priorities-no-offsets": dict(WDPM_PLAIN="0", WDPM_CLAMP="0", WDPM_PRIO="0", WDPM_DEM16="0"),
    "marching-everywhere-codes-skewed-heights": dict(WDPM_RELAY="0", WDPM_TRI="0", WDPM_DEM32="2", WDPM_BALANCE="2"),
    # round 5: the table with every slot of the resident round filled (chunks down to two row triples) and partner strips rounded half a triple apart
    "marching-everywhere-skewed-heights-slots-filled-paired": dict(WDPM_RELAY="0", WDPM_TRI="0", WDPM_BALANCE="2", WDPM_PAIR="2"),
    "relay-everywhere-four-waves": dict(WDPM_RELAY="2", WDPM_RELAY_NW="4", WDPM_RELAY_PRIO="2"),
    "relay-everywhere-eight-waves-codes": dict(WDPM_RELAY="2", WDPM_RELAY_NW="8", WDPM_DEM32="2", WDPM_RELAY_PRIO="2"),
    "triangle-everywhere-six-rows": dict(WDPM_TRI="2", WDPM_RELAY="0", WDPM_TRI_K="2"),
    "triangle-everywhere-three-rows-balance-off-no-graphs": dict(WDPM_TRI="2", WDPM_RELAY="0", WDPM_TRI_K="1", WDPM_BALANCE="0", WDPM_GRAPH="0"),
}


@pytest.mark.parametrize("name", list(VARIANTS))
def test_kernel_suites_with_the_variant_forced(name):
    env = dict(os.environ, **VARIANTS[name])
    p = subprocess.run([sys.executable, "-m", "pytest", *SUITE, "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1200)
    tail = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ""
    assert p.returncode == 0 and " passed" in tail and "failed" not in tail, f"{name} {VARIANTS[name]}:\n" + p.stdout[-3000:] + p.stderr[-1500:]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "forced_variants.txt"), "a") as f:
        f.write(f"{name} {VARIANTS[name]}: {tail}\n")
