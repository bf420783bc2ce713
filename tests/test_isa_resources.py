"""Static resources of the gfx950 kernels, read from the code objects the build leaves in gpd_amd/csrc/*.o (no GPU):
what DESIGN.md says about registers, LDS and scratch is what the compiler produced, and profiles/r06_isa_stats.txt is the
listing of the sources as committed."""
import importlib.util
import io
import os
from contextlib import redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def isa():
    from gpd_amd import api
    api.lib()  # builds the library (and with it the objects) when it is missing
    spec = importlib.util.spec_from_file_location("isa_stats", os.path.join(ROOT, "profiles", "isa_stats.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(os.path.join(mod.CSRC, "lenet_fast.o")):
        pytest.skip("no objects beside the library (prebuilt .so only)")
    return mod, {r["kernel"]: r for r in mod.collect()}


def test_scoring_kernels_keep_everything_in_registers(isa):
    _, k = isa
    names = [n for n in k if k[n]["unit"] in ("lenet", "lenet_fast")]
    assert len(names) >= 25
    for n in names:
        r = k[n]
        assert r["scratch"] == 0 and r["vgpr_spills"] == 0 and r["sgpr_spills"] == 0 and not r["dynamic_stack"], (n, r)
        assert r["vgpr"] + r["agpr"] <= 256, n  # two waves per SIMD at 512 threads: half of the 512 registers each
        assert r["lds"] <= 160 * 1024, n


def test_split_kernels_hold_the_matrix_instructions_the_bench_prices(isa):
    """bench.lenet_mfma_work multiplies tiles x instructions x operations per instruction: the instruction counts per
    tile / k-step are the ones in the code objects (nothing unrolled twice, nothing dropped)."""
    _, k = isa
    for C in (15, 12, 3, 1):
        ks = 3 if C <= 4 else (5 if C == 12 else 7)  # F1N_KS (four-byte pixels) / F1P_KS (twelve-byte pixels, round 6) / F1_KS k-steps x F1_MT row tiles per pixel tile
        assert k["gpd::conv1_i8_kernel<%d>" % C]["matrix"] == {"v_mfma_i32_16x16x64_i8": ks * 5}
        import bench
        assert bench.lenet_mfma_work(C)["conv1_i8_kernel"]["executed"] == 196 * ks * 5 * 32768.0
    assert k["gpd::conv2_bf16_kernel"]["matrix"] == {"v_mfma_f32_16x16x32_bf16": 16 * 6 + 8 * 6}  # k-steps x piece products per pixel tile and wave + the work unit of the wave that computes filters 48, 49 (2 conv rows x 4 k-steps)
    assert bench.lenet_mfma_work(15)["conv2_bf16_kernel"]["executed"] == (3 * 36 * 16 * 6 + 24 * 8 * 6) * 16384.0
    for nt in (1, 2, 3, 4, 5):
        assert k["gpd::fc1_bf16_kernel<%d>" % nt]["matrix"] == {"v_mfma_f32_16x16x32_bf16": 3 * nt * 4 * 6}  # m-tiles x n-tiles x pieces per BK; the step's body stands three times (the loop runs over pairs of steps, one stage each, + the last step)
    # LDS budgets of DESIGN.md section 4
    assert k["gpd::conv2_bf16_kernel"]["lds"] == 94080 + 62720 + 4
    assert k["gpd::conv1_i8_kernel<15>"]["lds"] <= 124 * 1024
    assert k["gpd::fc1_bf16_kernel<5>"]["lds"] == 159744


def test_default_geometry_kernels_do_not_touch_scratch(isa):
    _, k = isa
    for n in ("gpd::neighbourhood_kernel<false>", "gpd::neighbourhood_kernel<true>", "gpd::hand_eval_kernel", "gpd::plan_kernel",
              "gpd::shadow_set_kernel<0>", "gpd::shadow_image_kernel<6144, false>", "gpd::shadow_image_kernel<12288, false>",
              "gpd::grasp_image_kernel<false>", "gpd::normals_list_kernel", "gpd::normals_finish_kernel", "gpd::cluster_kernel"):
        assert k[n]["scratch"] == 0 and k[n]["vgpr_spills"] == 0, (n, k[n])


def test_committed_listing_is_of_the_committed_sources(isa):
    mod, _ = isa
    buf = io.StringIO()
    with redirect_stdout(buf):
        mod.main()
    want = open(os.path.join(ROOT, "profiles", "r06_isa_stats.txt")).read()
    assert buf.getvalue() == want, "kernels changed: python profiles/isa_stats.py > profiles/r06_isa_stats.txt"


def test_ip1_waits_for_its_lds_dma_before_every_barrier(isa):
    """fc1_bf16_kernel fills its LDS stages by global_load_lds_dwordx4: the data of a wave's DMA is ordered for the other waves'
    ds_reads only by that wave's vmcnt wait followed by the barrier.  The source relies on __syncthreads() emitting the wait
    (an LDS-DMA is a pending LDS write on the VM counter); this holds the compiler to it — every s_barrier of the kernel has
    an s_waitcnt vmcnt(0) right before it, and nothing stages through registers any more (no ds_write in the kernel)."""
    import subprocess
    import tempfile
    mod, _ = isa
    with tempfile.TemporaryDirectory() as tmp:
        co = mod.code_object("lenet_fast", tmp)
        dis = subprocess.run([mod.LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    body, on = [], False
    for line in dis.splitlines():
        if line[:1].isalnum() and line.rstrip().endswith(">:"):
            on = "fc1_bf16_kernelILi5" in line
            continue
        if on and line.startswith("\t"):
            body.append(line.split("//")[0].strip())
    assert any(l.startswith("global_load_lds_dwordx4") for l in body) and not any(l.startswith("ds_write") for l in body)
    barriers = [i for i, l in enumerate(body) if l.startswith("s_barrier")]
    assert len(barriers) >= 3
    for i in barriers:
        assert body[i - 1].startswith("s_waitcnt") and "vmcnt(0)" in body[i - 1], (i, body[i - 3:i + 1])
