"""PLY import with Scene::Load's conventions and image export.  CPU, plus one end-to-end test on
the GPU (PLY -> BVH -> render -> resolve -> PNG / PFM)."""
import os
import struct

import numpy as np
import pytest

CUBE_PLY = """ply
format ascii 1.0           { ascii/binary, format version number }
comment made by anonymous  { comments are keyword specified }
element vertex 8           { define "vertex" element, 8 in file }
property float32 x         { vertex contains float "x" coordinate }
property float32 y
property float32 z
element face 6             { there are 6 "face" elements in the file }
property list uint8 int32 vertex_index
                           { "vertex_indices" is a list of ints }
end_header                 { delimits the end of the header }
0 0 0                      { start of vertex list }
0 0 1
0 1 1
0 1 0
1 0 0
1 0 1
1 1 1
1 1 0
4 0 1 2 3                  { start of face list }
4 7 6 5 4
4 0 4 5 1
4 1 5 6 2
4 2 6 7 3
4 3 7 4 0
some trailing text
after the face list
"""


def test_ascii_ply_with_annotations_quads_and_trailing_text(hip, tmp_path):
    """the layout of the reference's Data/cube.ply: { } annotations in header AND data, quads, junk after the faces"""
    p = tmp_path / "cube.ply"
    p.write_text(CUBE_PLY)
    t = hip.load_ply(str(p))
    assert t.shape[0] == 12  # 6 quads -> fans of 2 (aiProcess_Triangulate, Scene.cpp:4-5)
    v = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 1], [0, 1, 0], [1, 0, 0], [1, 0, 1], [1, 1, 1], [1, 1, 0]], dtype=np.float32)
    # first quad 0 1 2 3 -> (0,1,2), (0,2,3); Triangle = {v0, v1 - v0, v2 - v0} (Scene.cpp:39-45); y/z swaps cancel
    assert np.array_equal(t["vert"][0], v[0]) and np.array_equal(t["e1"][0], v[1] - v[0]) and np.array_equal(t["e2"][0], v[2] - v[0])
    assert np.array_equal(t["vert"][1], v[0]) and np.array_equal(t["e1"][1], v[2] - v[0]) and np.array_equal(t["e2"][1], v[3] - v[0])
    assert np.all(t["materialType"] == 0)
    # the loaded mesh goes straight into the builder
    nodes, prims = hip.bvh_build(t)
    assert nodes[nodes["primitiveCount"] > 0]["primitiveCount"].sum() == 12


def test_binary_ply_with_extra_properties(hip, tmp_path):
    """binary little-endian, normals + uv after the position (the property set of Data/dragon.ply), triangles"""
    rng = np.random.default_rng(4)
    nv, nf = 50, 80
    verts = rng.uniform(-5, 5, size=(nv, 8)).astype(np.float32)
    faces = rng.integers(0, nv, size=(nf, 3)).astype(np.uint32)
    hdr = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % nv
    hdr += "".join(f"property float {n}\n" for n in ("x", "y", "z", "nx", "ny", "nz", "s", "t"))
    hdr += "element face %d\nproperty list uchar uint vertex_indices\nend_header\n" % nf
    p = tmp_path / "mesh.ply"
    with open(p, "wb") as f:
        f.write(hdr.encode())
        f.write(verts.tobytes())
        for tri in faces:
            f.write(struct.pack("<B3I", 3, *tri))
    t = hip.load_ply(str(p))
    assert t.shape[0] == nf
    assert np.array_equal(t["vert"], verts[faces[:, 0], :3])
    assert np.array_equal(t["e1"], verts[faces[:, 1], :3] - verts[faces[:, 0], :3])
    assert np.array_equal(t["e2"], verts[faces[:, 2], :3] - verts[faces[:, 0], :3])


def test_ply_errors(hip, tmp_path):
    with pytest.raises(hip.TyrError):
        hip.load_ply(str(tmp_path / "missing.ply"))
    bad = tmp_path / "bad.ply"
    bad.write_text("ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n3 0 1 2\n")
    with pytest.raises(hip.TyrError):  # face index out of range
        hip.load_ply(str(bad))
    notply = tmp_path / "x.ply"
    notply.write_text("solid\n")
    with pytest.raises(hip.TyrError):
        hip.load_ply(str(notply))


def test_reference_dragon_if_present(hip):
    """Data/dragon.ply of the reference checkout (only in the authoring container): 22,126 vertices, 37,986 triangles"""
    path = "/root/reference/PathTracer/Data/dragon.ply"
    if not os.path.exists(path):
        pytest.skip("reference data not present on this machine")
    t = hip.load_ply(path)
    assert t.shape[0] == 37986 and np.all(np.isfinite(t["vert"]))
    nodes, prims = hip.bvh_build(t)
    assert nodes[nodes["primitiveCount"] > 0]["primitiveCount"].sum() == 37986


def test_image_export(hip, tmp_path):
    W, H = 5, 3
    img = np.zeros((H * W, 4), dtype=np.float32)
    img[:, 0] = np.linspace(0, 1, H * W)
    img[:, 1] = 0.5
    img[3, 2] = np.nan  # a pixel without a completed path resolves to NaN (0/0, kernel.cu:658)
    hip.write_image(str(tmp_path / "a.ppm"), img, W, H)
    raw = (tmp_path / "a.ppm").read_bytes()
    assert raw.startswith(b"P6 5 3 255\n") and len(raw) == len(b"P6 5 3 255\n") + W * H * 3
    px = np.frombuffer(raw[len(b"P6 5 3 255\n"):], dtype=np.uint8).reshape(H * W, 3)
    assert px[0, 0] == 0 and px[-1, 0] == 255 and np.all(px[:, 1] == 128) and px[3, 2] == 0
    hip.write_image(str(tmp_path / "a.pfm"), img, W, H)
    raw = (tmp_path / "a.pfm").read_bytes()
    head = b"PF\n5 3\n-1.0\n"
    assert raw.startswith(head)
    data = np.frombuffer(raw[len(head):], dtype="<f4").reshape(H, W, 3)
    assert np.array_equal(data[::-1, :, 0].reshape(-1), img[:, 0])  # bottom-up rows
    # PNG: decoded independently (zlib + the PNG chunk layout), also for an image wider than one 64 KiB stored block
    import struct
    import zlib

    def decode_png(raw):
        assert raw[:8] == b"\x89PNG\r\n\x1a\n"
        pos, chunks = 8, []
        while pos < len(raw):
            n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
            body = raw[pos + 8:pos + 8 + n]
            assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + body)
            chunks.append((typ, body))
            pos += 12 + n
        assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
        w, h, depth, ctype = struct.unpack(">IIBB", chunks[0][1][:10])
        assert (depth, ctype) == (8, 2)
        rows = np.frombuffer(zlib.decompress(chunks[1][1]), dtype=np.uint8).reshape(h, w * 3 + 1)
        assert np.all(rows[:, 0] == 0)
        return rows[:, 1:].reshape(h * w, 3)

    hip.write_image(str(tmp_path / "a.png"), img, W, H)
    assert np.array_equal(decode_png((tmp_path / "a.png").read_bytes()), px)
    big = np.random.default_rng(1).random((40 * 700, 4)).astype(np.float32)
    hip.write_image(str(tmp_path / "b.png"), big, 700, 40)
    assert np.array_equal(decode_png((tmp_path / "b.png").read_bytes()), (np.clip(big[:, :3], 0, 1) * 255.0 + 0.5).astype(np.uint8))


def test_hostile_ply_headers_and_write_errors(hip, tmp_path):
    """tyr_load_ply trusts nothing in the header: counts are bounded by the file size before anything is sized from
    them, non-finite coordinates / indices / list counts are refused before any integer cast, and no exception leaves
    the extern "C" boundary; the image writers report a failed open or a short write as TYR_ERR_IO"""
    head = "ply\nformat ascii 1.0\nelement vertex {nv}\nproperty float x\nproperty float y\nproperty float z\nelement face {nf}\nproperty list uchar int vertex_indices\nend_header\n"
    body = "0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n"
    cases = {
        "huge_vertex_count": head.format(nv=2**62, nf=1) + body,       # would be a 55-exabyte resize
        "huge_face_count": head.format(nv=3, nf=2**40) + body,
        "nan_index": head.format(nv=3, nf=1) + "0 0 0\n1 0 0\n0 1 0\n3 0 nan 2\n",
        "nan_list_count": head.format(nv=3, nf=1) + "0 0 0\n1 0 0\n0 1 0\nnan 0 1 2\n",
        "inf_coordinate": head.format(nv=3, nf=1) + "0 0 0\ninf 0 0\n0 1 0\n3 0 1 2\n",
        "index_out_of_range": head.format(nv=3, nf=1) + "0 0 0\n1 0 0\n0 1 0\n3 0 1 3\n",
        "negative_index": head.format(nv=3, nf=1) + "0 0 0\n1 0 0\n0 1 0\n3 0 -1 2\n",
        "truncated": head.format(nv=3, nf=1) + "0 0 0\n1 0 0\n",
    }
    for name, text in cases.items():
        p = tmp_path / f"{name}.ply"
        p.write_text(text)
        with pytest.raises(hip.TyrError) as e:
            hip.load_ply(str(p))
        assert e.value.status in (-1, -5), (name, e.value.status)  # TYR_ERR_INVALID (or OOM), never a crash
    good = tmp_path / "good.ply"
    good.write_text(head.format(nv=3, nf=1) + body)
    assert hip.load_ply(str(good)).shape[0] == 1
    rgba = np.zeros((4, 4, 4), dtype=np.float32)
    for ext in ("ppm", "png", "pfm"):
        with pytest.raises(hip.TyrError) as e:
            hip.write_image(str(tmp_path / "no_such_dir" / f"x.{ext}"), rgba, 4, 4)
        assert e.value.status == -8  # TYR_ERR_IO
    if os.path.exists("/dev/full"):
        with pytest.raises(hip.TyrError) as e:
            hip.write_image("/dev/full", np.zeros((256, 256, 4), dtype=np.float32), 256, 256)  # ENOSPC on write / close
        assert e.value.status == -8


@pytest.mark.gpu
def test_ply_to_png_on_the_gpu(hip, orc, tmp_path):
    """SURVEY.md 8f-2 end to end on the GPU box: a mesh written as binary PLY -> tyr_load_ply (Scene::Load's conventions)
    -> tyr_bvh_build -> tyr_scene_upload -> tyr_render -> tyr_resolve -> tyr_write_png / tyr_write_pfm; the triangles that
    come out of the file are the generator's, the render equals the oracle's on the same arrays, the PFM holds the
    resolved frame's floats, the PNG parses"""
    import struct as st
    import zlib

    import torch

    from conftest import bits
    from tyrant_amd import scenes

    sc = scenes.mesh_scene(24)
    t = sc.triangles
    v = np.stack([t["vert"], t["vert"] + t["e1"], t["vert"] + t["e2"]], axis=1).reshape(-1, 3).astype(np.float32)
    keep = np.all((v.reshape(-1, 3, 3)[:, 1] - v.reshape(-1, 3, 3)[:, 0] == t["e1"]) & (v.reshape(-1, 3, 3)[:, 2] - v.reshape(-1, 3, 3)[:, 0] == t["e2"]), axis=1)
    assert keep.mean() > 0.9  # (vert + e) - vert == e for nearly every edge: those triangles must come back bit for bit
    nf = t.shape[0]
    p = tmp_path / "mesh.ply"
    with open(p, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nelement face %d\nproperty list uchar uint vertex_indices\nend_header\n" % (3 * nf, nf)).encode())
        f.write(v.tobytes())
        for i in range(nf):
            f.write(st.pack("<BIII", 3, 3 * i, 3 * i + 1, 3 * i + 2))
    loaded = hip.load_ply(str(p))
    assert loaded.shape[0] == nf
    for fld in ("vert", "e1", "e2"):
        assert np.array_equal(bits(loaded[fld][keep]), bits(t[fld][keep])), fld
    nodes, prims = hip.bvh_build(loaded)
    W, H, N, spp = 160, 96, 8192, 2
    g = hip.Renderer(W, H, N)
    g.upload(nodes, prims), g.set_spheres(sc.spheres), g.set_camera(sc.camera), g.set_sun_position(*sc.sun_position)
    o = orc.Oracle(W, H, N)
    o.upload(nodes, prims), o.set_spheres(sc.spheres), o.set_camera(sc.camera), o.set_sun_position(*sc.sun_position)
    assert g.render(spp) == o.render(spp)
    bg, bo = g.blit_buffer(), o.blit_buffer()
    assert np.array_equal(bg[:, 3], bo[:, 3]) and np.allclose(bg[:, :3], bo[:, :3], rtol=1e-5, atol=1e-6)
    out = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    g.resolve_into(out.data_ptr())
    frame = out.cpu().numpy().reshape(H * W, 4)
    pfm, png = tmp_path / "frame.pfm", tmp_path / "frame.png"
    hip.write_image(str(pfm), frame, W, H)
    hip.write_image(str(png), frame, W, H)
    raw = open(pfm, "rb").read()
    head, body = raw.split(b"-1.0\n", 1)
    assert head == b"PF\n%d %d\n" % (W, H)
    got = np.frombuffer(body, dtype="<f4").reshape(H, W, 3)[::-1]  # PFM rows run bottom to top
    assert np.array_equal(bits(got), bits(frame.reshape(H, W, 4)[:, :, :3]))
    data = open(png, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n" and st.unpack(">II", data[16:24]) == (W, H)
    idat = data[data.index(b"IDAT") + 4 : data.index(b"IEND") - 8]
    rows = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(H, 1 + 3 * W)
    c = np.nan_to_num(frame.reshape(H, W, 4)[:, :, :3], nan=0.0).clip(0.0, 1.0)
    assert np.all(rows[:, 0] == 0) and np.array_equal(rows[:, 1:].reshape(H, W, 3), (c * np.float32(255.0) + np.float32(0.5)).astype(np.uint8))
