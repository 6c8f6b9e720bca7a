"""CPU-side checks of the boundary: the shared library loads and exports every function
include/rt_mi355.h declares, the headers are valid C and C++, wire-struct sizes agree between the
headers, the ctypes views and numpy dtypes, host-side helpers behave, and computing entry points
fail loudly (never fall back) when no GPU is usable."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "rt_mi355.h")


def _declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(rt):
    from ray_tracer_amd import _lib
    L = _lib.load()
    names = _declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in rt_mi355.h but not exported"
    assert set(names) == set(_lib.PROTOTYPES), set(names) ^ set(_lib.PROTOTYPES)


def test_headers_compile_as_c_and_cpp():
    prog = '#include "rt_mi355.h"\nint main(void){rt_params p; (void)p; return (int)sizeof(rt_hit) - 8;}\n'
    with tempfile.TemporaryDirectory() as d:
        for ext, cc, std in ((".c", "gcc", "-std=c11"), (".cpp", "g++", "-std=c++17")):
            path = os.path.join(d, "t" + ext)
            open(path, "w").write(prog)
            exe = os.path.join(d, "t" + ext + ".out")
            subprocess.check_call([cc, std, "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), path, "-o", exe])
            assert subprocess.call([exe]) == 0


def test_wire_struct_sizes(rt):
    # sizes asserted in include/rt_types.h (reference src/types.rs, src/grid.rs)
    assert C.sizeof(rt.CameraUniform) == 80 and C.sizeof(rt.GridMetadata) == 96 and C.sizeof(rt.RayDebugInfo) == 80
    assert rt.BOX_DTYPE.itemsize == 96 and rt.TRIANGLE_DTYPE.itemsize == 80 and rt.HIT_DTYPE.itemsize == 8
    assert C.sizeof(rt.RtStats) == 88 and C.sizeof(rt.RtShard) == 24 and C.sizeof(rt.RtParams) == 68


def test_size_table_of_the_header_matches_every_binding(rt):
    """RT_SIZEOF_* at the end of rt_mi355.h is what bindings in other languages assert (ffi/src/lib.rs cannot be compiled
    here): the C and C++ compilers check it against the header's structs (test_headers_compile_as_c_and_cpp), this test
    checks the ctypes views and the Rust source's const asserts against the same numbers."""
    from ray_tracer_amd import _lib
    table = {k: int(v) for k, v in re.findall(r"#define RT_SIZEOF_([A-Z_]+) (\d+)", open(HDR).read())}
    views = {"PARAMS": rt.RtParams, "SHARD": rt.RtShard, "STATS": rt.RtStats, "RENDER_OPTS": _lib.RtRenderOpts, "TEXTURE": _lib.RtTexture,
             "BVH_STATS": _lib.RtBvhStats}
    assert set(views) | {"HIT"} == set(table)
    for name, view in views.items():
        assert C.sizeof(view) == table[name], name
    assert rt.HIT_DTYPE.itemsize == table["HIT"]
    rust = open(os.path.join(ROOT, "ffi", "src", "lib.rs")).read()
    asserted = {k: int(v) for k, v in re.findall(r"size_of::<Rt(\w+)>\(\) == (\d+)", rust)}
    assert asserted == {"Params": table["PARAMS"], "Shard": table["SHARD"], "Hit": table["HIT"]}
    fields = re.search(r"pub struct RtParams \{(.*?)\n\}", rust, re.S).group(1)
    n_words = sum(3 if "[f32; 3]" in f else 1 for f in fields.split(",") if ":" in f)
    assert 4 * n_words == table["PARAMS"], "ffi/src/lib.rs RtParams has a different number of fields"


def test_oracle_binding_shares_the_header_structs(rt, ob):
    """The oracle is compiled against the same rt_params / rt_stats / rt_shard as the library (oracle.h includes the
    public header for its types), so its Python binding has to lay them out the same way, field for field."""
    for mine, theirs in ((ob.Params, rt.RtParams), (ob.Stats, rt.RtStats), (ob.Shard, rt.RtShard), (ob.CameraUniform, rt.CameraUniform)):
        assert C.sizeof(mine) == C.sizeof(theirs), mine
        assert [(n, getattr(mine, n).offset) for n, _ in mine._fields_] == [(n, getattr(theirs, n).offset) for n, _ in theirs._fields_], mine


def test_default_params_are_the_reference_constants(rt):
    p = rt.default_params()
    assert (p.finest_cell_size, p.max_objects_per_cell, p.max_dda_steps) == (16.0, 8192, 200)  # grid.rs:6-7, wgsl:520
    assert (p.wgsl_compat, p.lod_cull, p.moving_box_count) == (0, 1, 3)
    assert abs(p.fov - 0.785398) < 1e-7 and list(p.light_direction) == [0.5, -1.0, pytest.approx(0.3)]
    assert p.shadow_offset == pytest.approx(0.001)


def test_camera_matches_oracle_bitwise(rt, ob):
    # src/camera.rs:59-66 start poses, :76-122 basis and uniform
    for scene in ("composed", "walls", "tunnel", "gltf", "pyramid", "fractal"):
        cam = rt.Camera.for_scene(scene)
        for h in (36.0, 1080.0):
            a = cam.to_uniform(1.25, h, rt.DEFAULT_FOV, True)
            b = ob.camera_to_uniform(cam.position, cam.yaw, cam.pitch, 1.25, h, rt.DEFAULT_FOV, True)
            assert bytes(a) == bytes(b)
    u = rt.Camera.for_scene("gltf").to_uniform(0.0, 1080.0)
    assert u.min_pixel_size == 2.0 and abs(u.lod_factor - 1080.0 / (2 * np.tan(0.785398 / 2))) < 1e-2
    assert list(u.up) == [0.0, 1.0, 0.0]


def test_shard_pixels_matches_oracle(rt, ob):
    for (w, h, tw, th, n) in ((1920, 1080, 64, 64, 8), (70, 45, 16, 8, 3), (8, 8, 64, 64, 2), (100, 3, 7, 2, 5)):
        for r in range(n):
            assert rt.shard_pixels(w, h, rt.RtShard(tw, th, r, n)) == ob.shard_pixels(w, h, ob.Shard(tw, th, r, n))
    assert rt.shard_pixels(33, 7, None) == 231
    # unequal shares through an owner table; together the ranks cover every tile once
    from ray_tracer_amd import sharding
    assert sharding.owner_table([3, 4, 4]) == [0, 1, 2, 0, 1, 2, 0, 1, 2, 1, 2]
    # what bench.py deals at N = 2, 4, 8: the gathering rank three parts in 64 short of an equal share
    assert sharding.gatherer_weights(1) is None and sharding.gatherer_weights(2) == [29, 32]
    assert sharding.gatherer_weights(4) == [13, 16, 16, 16] and sharding.gatherer_weights(8) == [5] + [8] * 7
    for (w, h, tw, th, weights) in ((1920, 1080, 32, 32, (5, 8, 8, 8, 8, 8, 8, 8)), (150, 85, 32, 16, (2, 3, 3)), (70, 45, 16, 8, (1, 5))):
        owner = sharding.owner_table(weights)
        assert len(owner) == sum(weights) and all(owner.count(r) == k for r, k in enumerate(weights))
        total = 0
        for r in range(len(weights)):
            a = rt.shard_pixels(w, h, rt.RtShard(tw, th, r, owner=owner))
            assert a == ob.shard_pixels(w, h, ob.Shard(tw, th, r, owner=owner))
            total += a
        assert total == -(-w // tw) * -(-h // th) * tw * th


def test_no_gpu_means_error_not_fallback(rt):
    """On a machine without a gfx950 device rt_create must fail; on the GPU box it must succeed."""
    import torch
    if torch.cuda.is_available():
        rt.TraceCore(0).close()
    else:
        with pytest.raises(rt.RtError) as e:
            rt.TraceCore(0)
        assert e.value.code == -2  # RT_ERR_NO_DEVICE


def test_product_package_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "ray-tracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), f"{f} mentions the oracle"


def test_missing_rccl_is_an_error_code_not_a_crash():
    """Without an RCCL to bind the multi-GPU entry points return RT_ERR_COMM (include/rt_mi355.h): the process must
    survive to take another route. RT_RCCL_LIBRARY names the one library to bind; the binding is cached per process,
    hence the child."""
    prog = ("import ctypes as C, sys; sys.path.insert(0, %r)\n"
            "from ray_tracer_amd import _lib\n"
            "L = _lib.load(); buf = (C.c_uint8 * _lib.RT_COMM_ID_BYTES)()\n"
            "print(L.rt_comm_unique_id(buf), L.rt_comm_unique_id(buf))\n" % ROOT)
    env = dict(os.environ, RT_RCCL_LIBRARY="/nonexistent/librccl-none.so")
    out = subprocess.run([os.sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["-6", "-6"], out.stdout
