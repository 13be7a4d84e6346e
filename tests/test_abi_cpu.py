"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol the header declares,
fails loudly without a GPU, and the exporter writes what the loader expects. No compute on the device."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(pkg):
    from unina_yolo_dla_amd import build, engine
    build.build_native()
    return engine.load_library()


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "unina_mi355.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{]*\)\s*;", text)
    return sorted(set(n for n in names if n.startswith("unina_") or n in (
        "init_postprocess_resources", "cleanup_postprocess_resources", "reset_detection_counter",
        "get_detection_count", "decode_yolo_head", "run_gpu_nms", "copy_valid_detections_to_host",
        "create_norm_params_imagenet", "create_norm_params", "preprocess_bgra_resize", "preprocess_bgra",
        "preprocess_nv12", "allocate_preprocess_buffer", "free_preprocess_buffer", "create_preprocess_stream",
        "destroy_preprocess_stream")))


def test_header_and_library_agree(lib):
    from unina_yolo_dla_amd import engine
    declared = _declared_functions()
    assert len(declared) >= 22
    assert sorted(engine.ABI_SYMBOLS) == declared          # the Python binding list mirrors the header
    for name in declared:
        assert hasattr(lib, name), f"libunina_mi355.so does not export {name}"
    assert b"gfx950" in lib.unina_version()
    from unina_yolo_dla_amd import build
    assert lib.unina_version().endswith(b"src:" + build.source_hash().encode())   # built from THIS tree (build.py hashes csrc/ + include/)


def test_reference_postprocess_symbols_present(lib):
    # the seven entry points of gpu_postprocess.h:42-80, by the reference's exact names
    for name in ("init_postprocess_resources", "cleanup_postprocess_resources", "reset_detection_counter",
                 "get_detection_count", "decode_yolo_head", "run_gpu_nms", "copy_valid_detections_to_host"):
        assert hasattr(lib, name)


def test_detection_record_layout(pkg):
    from unina_yolo_dla_amd import engine
    d = engine.DET_DTYPE
    assert d.itemsize == 32                                  # gpu_postprocess.h:27-33, __align__(32)
    assert [d.fields[n][1] for n in ("x1", "y1", "x2", "y2", "confidence", "class_id", "valid", "_pad")] == \
        [0, 4, 8, 12, 16, 20, 24, 28]


def test_load_errors_are_loud_not_fatal(lib, tmp_path):
    h = C.c_void_p()
    assert lib.unina_load_engine(b"/nonexistent/engine.une", 0, C.byref(h)) == 1          # UNINA_ERR_IO
    assert b"cannot open" in lib.unina_last_error(None)
    bad = tmp_path / "bad.une"
    bad.write_bytes(b"NOTANENG" + b"\0" * 200)
    assert lib.unina_load_engine(str(bad).encode(), 0, C.byref(h)) == 2                     # UNINA_ERR_FORMAT
    assert b"magic" in lib.unina_last_error(None)
    assert not h.value
    lib.unina_unload_engine(None)                                                            # NULL is a no-op


def test_engine_wrapper_refuses_without_gpu(pkg, sd7, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from unina_yolo_dla_amd import engine, export
    path = str(tmp_path / "a.une")
    export.export_engine(sd7, path)
    with pytest.raises(engine.EngineError):
        engine.Engine(path)


def test_export_tables(pkg, sd7, tmp_path):
    from unina_yolo_dla_amd import export
    b = export.EngineBuilder(sd7)
    assert len(b.ops) == 52                                   # 67 reference convs -> 50 conv launches + stem + pool
    convs = [s.module for o in b.ops for s in o.segs if o.kind in (export.OP_CONV, export.OP_STEM)]
    assert sorted(convs) == sorted(n.name for n in pkg.graph.Graph().convs())   # every conv exactly once
    path = str(tmp_path / "a.une")
    b.save(path)
    h = export.read_engine_header(path)
    assert (h["in_h"], h["in_w"], h["num_classes"], h["n_ops"]) == (640, 640, 4, 52)
    assert h["macs"] == 17_832_345_600
    assert os.path.getsize(path) == 128 + 64 * h["n_buffers"] + 256 * h["n_ops"] + h["blob_bytes"]


def test_bn_folding_matches_unfolded_math(pkg, sd7):
    from unina_yolo_dla_amd import export
    w, b = export.fold_bn(sd7, "backbone.stage1_conv")
    x = np.random.default_rng(0).standard_normal((64,)).astype(np.float64)        # conv output for one pixel
    g, beta = sd7["backbone.stage1_conv.bn.weight"], sd7["backbone.stage1_conv.bn.bias"]
    mu, var = sd7["backbone.stage1_conv.bn.running_mean"], sd7["backbone.stage1_conv.bn.running_var"]
    unfolded = (x - mu) / np.sqrt(var.astype(np.float64) + 1e-5) * g + beta                # model.py:46-50
    scale = w[:, 0, 0, 0] / sd7["backbone.stage1_conv.conv.weight"][:, 0, 0, 0].astype(np.float64)
    np.testing.assert_allclose(x * scale + b, unfolded, rtol=1e-12, atol=1e-12)


def test_export_embeds_narrow_widths(pkg):
    """base_channels=16 (model.py:331-333) is below the kernels' 32-channel K block: the exporter embeds it at width
    32 (numerics: test_op_table_cpu.py::test_narrow_base_channels_are_embedded_exactly)."""
    from unina_yolo_dla_amd import export
    g = pkg.graph.Graph(base_channels=16, in_h=64, in_w=64)
    b = export.EngineBuilder(pkg.synth.make_state_dict(7, g), g)
    assert b.narrow_base_channels == 16 and b.g.base_channels == 32
    assert all(op.cin % 32 == 0 for op in b.ops if op.kind == export.OP_CONV)


def test_c3k2_groups_are_recognised_at_load(lib, pkg, sd7, tmp_path):
    """Load-time block fusion (csrc/c3k2_fused.hip, head_fused.hip): the structural matcher finds the seven C3k2 blocks
    of graph (A) (model.py:76-110) and the P2 DetectionHead (model.py:274-303; the only head whose weights a workgroup
    can stream) in the exporter's op table, one block fewer in the lite_p2 variant, none in an fp32 engine file."""
    from unina_yolo_dla_amd import export
    g = pkg.graph.Graph(in_h=64, in_w=64)
    path = str(tmp_path / "a.une")
    export.export_engine(sd7, path, g)
    assert lib.unina_debug_fusable_groups(path.encode()) == 9      # 7 C3k2 blocks + P2 head + the sppf.cv2 -> lateral_p3 pair
    export.export_engine(sd7, path, g, precision=export.FP32)
    assert lib.unina_debug_fusable_groups(path.encode()) == 0
    gl = pkg.graph.Graph(in_h=64, in_w=64, lite_p2=True)
    export.export_engine(pkg.synth.make_state_dict(7, gl), path, gl)
    assert lib.unina_debug_fusable_groups(path.encode()) == 8
    assert lib.unina_debug_fusable_groups(b"/nonexistent.une") == -1
    # INT8 engine file: the five blocks whose tensors are all int8 (stage2/3, fpn_c3k2_1, pan_c3k2_1/2) as int8 block
    # kernels, the two h = 32 blocks (fp16 by builder choice) and the carved-out P2 head (train.py:779) as fp16 ones
    from emulate import run_op_table
    b16 = export.EngineBuilder(sd7, g)
    amax = export.calibrate(run_op_table(b16, pkg.rng.frame(5000 + i, 64, 64))[1] for i in range(2))
    export.export_engine(sd7, path, g, precision=export.INT8, amax=amax)
    assert lib.unina_debug_fusable_groups(path.encode()) == 9


def test_header_is_plain_c_and_record_is_32_bytes(tmp_path):
    """include/unina_mi355.h must be consumable by a C compiler without the HIP headers (cgo / ctypes-gen / a C node):
    gcc -std=c99 -pedantic on the header itself, then a C translation unit that checks the record layout the reference
    fixes (gpu_postprocess.h:27-33) at compile time."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "unina_mi355.h")
    r = subprocess.run(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Wpedantic", "-Werror",
                        "-DUNINA_NO_HIP_HEADERS", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    src = tmp_path / "layout.c"
    src.write_text('#define UNINA_NO_HIP_HEADERS\n#include "unina_mi355.h"\n#include <stddef.h>\n'
                   'typedef char rec32[(sizeof(GpuDetection) == 32) ? 1 : -1];\n'
                   'typedef char conf16[(offsetof(GpuDetection, confidence) == 16) ? 1 : -1];\n'
                   'typedef char valid24[(offsetof(GpuDetection, valid) == 24) ? 1 : -1];\n'
                   'int use(unina_engine_t *e) { int w, h, n; return unina_engine_input_dims(e, &w, &h, &n); }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                        "-o", str(tmp_path / "layout.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_node_harness_builds_and_links_against_the_library(pkg):
    """tools/node_harness.cpp (the reference node's per-frame body, perception_node.cpp:581-689, over the header alone)
    compiles with hipcc and resolves every symbol it uses from libunina_mi355.so; without arguments it prints its usage
    before touching the GPU."""
    import subprocess
    from unina_yolo_dla_amd import build
    build.build_native()
    exe = build.build_harness()
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage:" in r.stderr
    nm = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    for sym in ("unina_load_engine", "unina_enqueue", "unina_infer", "unina_infer_bgra", "decode_yolo_head", "run_gpu_nms",
                "copy_valid_detections_to_host", "preprocess_bgra_resize", "reset_detection_counter", "get_detection_count"):
        assert sym in nm, sym


def test_comm_entry_points_reject_bad_arguments_without_touching_a_gpu(lib):
    """include/unina_mi355.h "multi-GPU": argument errors of the RCCL gather entry points are reported through the return
    code and unina_comm_last_error before any HIP / RCCL call is made (so this runs on the CPU-only container too)."""
    import ctypes as C
    comm = C.c_void_p()
    ident = (C.c_char * 128)()
    assert lib.unina_comm_init(C.byref(comm), ident, 2, 1, 0) != 0 and not comm.value          # rank outside the world
    assert b"bad arguments" in lib.unina_comm_last_error()
    assert lib.unina_comm_init(None, ident, 0, 1, 0) != 0
    assert lib.unina_comm_all_gather(None, None, None, 0, None) != 0
    assert lib.unina_comm_unique_id(None) != 0
    assert lib.unina_comm_rank(None) == -1 and lib.unina_comm_world(None) == -1
    lib.unina_comm_destroy(None)                                                                 # a no-op
