"""the reference-side adapter (integration/hip/HIPBackend.{hpp,cpp}, HIPOps.cpp: mllm's Backend / Op registry on the C ABI) is real code.
Where the reference tree exists (this container; never the GPU box) it is compiled against the reference's own headers and linked, with --no-undefined,
against the compiled reference library and libmllm_hip.so; the test then checks that the objects the registry needs are in it.  Test infrastructure only:
nothing reference-built enters the product path."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/mllm"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not present (GPU box)")
def test_hip_adapter_compiles_and_links_against_the_reference():
    from mllm_amd import build as b
    b.build()
    subprocess.run(["make", "-f", "oracle/Makefile.ref", "-j8", "adapter"], cwd=ROOT, check=True, capture_output=True, timeout=1500)
    so = os.path.join(ROOT, "oracle", "_ref", "libmllm_hip_adapter.so")
    assert os.path.exists(so)
    syms = subprocess.run(["nm", "-DC", so], capture_output=True, text=True, check=True).stdout
    for want in ("mllm::HIPBackend::runOp", "mllm::HIPBackend::runForward", "mllm::HIPBackend::opCreate", "mllm::HIPBackend::load_from_file",
                 "mllm::HIPBackend::alloc_device", "mllm::HIPBackend::registerOps", "mllm::registerHIPBackendCreator", "vtable for mllm::HIPBackend"):
        assert want in syms, want
    # every call out of the adapter into the product goes through the C ABI: undefined symbols are either the reference's (mllm::...) / libstdc++ / libc
    # or mllm_hip_* entry points that include/mllm_hip.h declares
    from mllm_amd import lib
    declared = set(lib.declared_symbols())
    used = {l.split()[-1] for l in syms.splitlines() if " U mllm_hip_" in l}
    assert used and used <= declared, used - declared
    assert {"mllm_hip_linear_q4kp_packed", "mllm_hip_fa2", "mllm_hip_rope_apply", "mllm_hip_rmsnorm", "mllm_hip_embedding_q40", "mllm_hip_patch_gemm_f32"} <= used


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not present (GPU box)")
def test_reference_module_runs_through_the_adapter_on_the_null_device(tmp_path):
    """The adapter's plumbing, executed in the container: the reference's own Qwen2VLModel (Module / Layer / Tensor frontend compiled from the reference tree) is moved
    onto the HIP backend and run -- trace pass of `to()`, load pass, image prefill, decode steps -- with the device entry points of the C ABI replaced by the null device of
    oracle/mock/mock_hip_abi.cpp (host memory; every launcher reads all its inputs and writes all its outputs) under AddressSanitizer.  No arithmetic is checked here (that
    is tests/test_gpu_adapter.py on the MI355X); what is checked: no Op was refused (no CPU fallback), no ASan report, and device blocks do not leak from step to step."""
    import json

    import numpy as np

    from mllm_amd import build as b, synth
    from mllm_amd import synthfile as weights
    b.build()
    subprocess.run(["make", "-f", "oracle/Makefile.ref", "-j8", "mock"], cwd=ROOT, check=True, capture_output=True, timeout=1500)
    exe = os.path.join(ROOT, "oracle", "_ref", "mock_hip_qwen2vl")
    c = synth.qwen2vl_tiny()
    path = weights.qwen2vl_file(c)
    pix, grid, ids = synth.qwen2vl_inputs(c, (8, 8), 6)
    pix.tofile(str(tmp_path / "pix.f32"))
    ids.tofile(str(tmp_path / "ids.i32"))
    cfg = f"{c.hidden},{c.inter},{c.layers},{c.heads},{c.kv_heads},{c.vocab},{c.v_dim},{c.cache_limit},{c.image_token_id},{c.vision_start_token_id},{c.vision_end_token_id},{c.video_token_id}"
    reports = []
    for steps in (2, 9):
        out = subprocess.run([exe, "--model", path, "--ids", str(tmp_path / "ids.i32"), "--pix", str(tmp_path / "pix.f32"), "--grid", "1,8,8", "--steps", str(steps), "--threads", "2",
                              "--out", str(tmp_path), "--cfg", cfg], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert out.returncode == 0 and "AddressSanitizer" not in out.stderr, (out.returncode, out.stderr[-3000:])
        reports.append(json.loads(next(l for l in out.stdout.splitlines() if l.startswith('{"backend"'))))
    for r in reports:
        assert r["cpu_fallback_ops"] == 0 and r["refused"] == [], r
    assert reports[0]["live_device_blocks"] == reports[1]["live_device_blocks"], reports      # nothing accumulates over decode steps
    assert reports[1]["hip_ops_run"] > reports[0]["hip_ops_run"] > 600
    assert len(np.fromfile(str(tmp_path / "tokens.i32"), dtype=np.int32)) == 9


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not present (GPU box)")
def test_other_configs_modules_run_through_the_adapter_on_the_null_device(tmp_path):
    """The same for the other four BASELINE configs' Modules (QWenForCausalLM, TinyLLaMAModel, ViTModel, the LLaVA graph; oracle/ref_drivers/ref_hip_{llm,vit,llava}.cpp):
    trace pass, load pass, prefill and decode steps on the null device under AddressSanitizer -- no Op refused (F_CAT, F_FLATTEN, the two-pair F_TRANPOSE, the fp32
    EMBEDDING and F_INDEX_PUT(accumulate) all have creators), no ASan report."""
    import json

    from mllm_amd import build as b, mllmfile as mf, synth
    from mllm_amd import synthfile as weights
    b.build()
    subprocess.run(["make", "-f", "oracle/Makefile.ref", "-j8", "mock"], cwd=ROOT, check=True, capture_output=True, timeout=1500)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    td = str(tmp_path)

    def run(cmd):
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0 and "AddressSanitizer" not in out.stderr, (cmd[0], out.returncode, out.stderr[-3000:])
        r = json.loads(next(l for l in out.stdout.splitlines() if l.startswith('{"backend"')))
        assert r["cpu_fallback_ops"] == 0 and r["refused"] == [], r
        return r

    for fam, c in (("qwen", synth.qwen15_tiny()), ("tinyllama", synth.tinyllama_tiny(mf.Q4_K))):
        synth.causal_lm_ids(c, 20).tofile(os.path.join(td, "ids.i32"))
        cfg = f"{c.hidden},{c.inter},{c.layers},{c.heads},{c.kv_heads},{c.vocab},{c.cache_limit},{int(c.tie_embedding)}"
        r = run([os.path.join(ROOT, "oracle", "_ref", "mock_hip_llm"), "--family", fam, "--model", weights.causal_lm_file(c), "--ids", os.path.join(td, "ids.i32"), "--steps", "4",
                 "--threads", "2", "--out", td, "--cfg", cfg])
        assert r["hip_ops_run"] > 150
    c = synth.vit_tiny()
    synth.vit_images(c, 2).tofile(os.path.join(td, "img.f32"))
    r = run([os.path.join(ROOT, "oracle", "_ref", "mock_hip_vit"), "--model", weights.vit_file(c), "--img", os.path.join(td, "img.f32"), "--n", "2", "--threads", "2", "--out", td,
             "--cfg", f"{c.hidden},{c.heads},{c.ffn},{c.blocks},{c.patch},{c.img},{c.classes}"])
    assert r["hip_ops_run"] > 60
    c = synth.llava_tiny()
    ids, img = synth.llava_inputs(c)
    ids.tofile(os.path.join(td, "ids.i32"))
    img.tofile(os.path.join(td, "img.f32"))
    cfg = f"{c.hidden},{c.heads},{c.inter},{c.layers},{c.vocab},{c.cache_limit},{c.v_hidden},{c.v_heads},{c.v_ffn},{c.v_blocks},{c.patch},{c.img}"
    base = [os.path.join(ROOT, "oracle", "_ref", "mock_hip_llava"), "--model", weights.llava_file(c), "--ids", os.path.join(td, "ids.i32"), "--img", os.path.join(td, "img.f32"),
            "--threads", "2", "--out", td, "--cfg", cfg]
    assert run(base + ["--steps", "4"])["hip_ops_run"] > 200
    run(base + ["--dump-vision", "1"])


def test_integration_doc_lists_every_registered_creator():
    """INTEGRATION.md §4 is the adapter's registerOps(), row for row: every `creators_[KEY]` of integration/hip/HIPOps.cpp is named in §4, every C-ABI entry point an
    Op class calls is named there too, and §4 names no HIP*Op class that does not exist."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "integration", "hip", "HIPOps.cpp")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 4. `registerOps()`"):doc.index("## 5. Build")]
    keys = re.findall(r"creators_\[([A-Z0-9_]+)\]", src)
    assert len(keys) >= 25
    for k in keys:
        assert f"`{k}`" in sec, f"creator {k} is registered but INTEGRATION.md §4 does not list it"
    classes = set(re.findall(r"^class (HIP[A-Za-z0-9]+Op) ", src, flags=re.M)) - {"HIPOp"}
    for c in re.findall(r"`(HIP[A-Za-z0-9]+Op)`", sec):
        assert c in classes, f"INTEGRATION.md §4 names {c}, which integration/hip/HIPOps.cpp does not define"
    for c in classes:
        assert f"`{c}`" in sec, f"{c} is defined but INTEGRATION.md §4 does not describe it"
    body = src[:src.index("void HIPBackend::registerOps")]
    for fn in sorted(set(re.findall(r"\b(mllm_hip_[a-z0-9_]+)\(", body))):
        if fn in ("mllm_hip_last_error", "mllm_hip_sync"):
            continue
        assert fn in doc, f"{fn} is called by the adapter but INTEGRATION.md does not mention it"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not present (GPU box)")
def test_lazy_window_runs_and_aliasing_rule_on_the_null_device():
    """The adapter's lazy window (HIPBackend::lazy / emit_group) against hand-made runs of Ops (oracle/ref_drivers/ref_hip_window.cpp on the null device, under
    AddressSanitizer): the runs it folds into one launch, and -- the part a race would hide in -- the aliasing patterns the frontend's block re-use produces: an output may share
    memory with an input of the same launch only where one workgroup owns the slice on both sides; otherwise the run goes out Op by Op, in program order."""
    import json

    from mllm_amd import build as b
    b.build()
    subprocess.run(["make", "-f", "oracle/Makefile.ref", "-j8", "mock"], cwd=ROOT, check=True, capture_output=True, timeout=1500)
    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "mock_hip_window")], capture_output=True, text=True, timeout=300, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0 and "AddressSanitizer" not in out.stderr, (out.returncode, out.stderr[-3000:])
    got = {r["name"]: (r["calls"], r["fused_launches"], r["fused_ops"]) for r in json.loads(out.stdout.strip().splitlines()[-1])}
    want = {
        "NLLL": (1, 1, 4), "LA": (1, 1, 2), "LA_other_order": (1, 1, 2), "NLSLM": (1, 1, 5), "ANLLL": (1, 1, 5), "RRKKF_gqa": (1, 1, 5), "RRKKF_mha": (1, 1, 5),
        "NLSLM_up_reuses_gate_block": (1, 1, 5),            # outputs that alias each other: written by one thread in the Ops' order
        "RRKKF_out_over_q_gqa": (1, 1, 5), "RRKKF_out_over_q_mha": (1, 1, 5), "RRKKF_krot_over_q_mha": (1, 1, 5),      # the same workgroup owns the slice on both sides
        "NLLL_output_over_input": (2, 1, 3),                # the norm alone, then q | k | v on its output (the aliased block is no input of THAT launch)
        "LA_sum_over_input_row": (2, 0, 0),                 # other workgroups still read the row
        "ANLLL_norm_output_over_add_operand": (2, 1, 4),    # the add alone, then norm + q | k | v (LLaVA-7B's case)
        "RRKKF_krot_over_q_gqa": (5, 0, 0),                 # six query heads per group read what one workgroup would overwrite
        "NLSLM_broken_chain": (4, 1, 2), "L_unfusable_then_A": (2, 0, 0), "RRKKF_wrong_row_gqa": (2, 1, 4), "RRKKF_wrong_row_mha": (2, 1, 4), "flush_before_other": (2, 0, 0),
    }
    assert got == want, {k: (got.get(k), want[k]) for k in want if got.get(k) != want[k]}
