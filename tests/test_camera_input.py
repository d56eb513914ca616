"""SURVEY.md 8f-4, the headless form: Camera::handle_input (camera.cpp:3-44) as a pure function of the key / cursor
state the reference reads from its GLFW window, and the PERFORMANCE_TEST fly-through (performance_measure.cpp:7-45).

CPU: tyr_camera_handle_input (host code of the library) against the oracle's restatement, bit for bit over random
states, and against hand-computed cases.  GPU: examples/flythrough.cpp renders the three recorded views and writes
Performance.txt in the reference's layout; the accumulation resets when the camera moves and only then."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def poses(hip, orc, position, direction, up, h, v):
    a = hip.CameraPose((C.c_float * 3)(*position), (C.c_float * 3)(*direction), (C.c_float * 3)(*up), h, v)
    b = orc.CameraPose((C.c_float * 3)(*position), (C.c_float * 3)(*direction), (C.c_float * 3)(*up), h, v)
    return a, b


def same(a, b):
    return bytes(a.position) == bytes(b.position) and a.horizontal_angle == b.horizontal_angle and a.vertical_angle == b.vertical_angle


def test_handle_input_matches_oracle_bit_for_bit(hip, orc):
    rng = np.random.default_rng(11)
    L = orc.lib()
    for _ in range(4000):
        keys = [int(x) for x in rng.integers(0, 2, 8)]
        cur = (float(rng.uniform(0, 1920)), float(rng.uniform(0, 1080)))
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        pos = rng.uniform(-200, 200, 3)
        h, v = float(rng.uniform(-20000, 20000)), float(rng.uniform(-1.7, 1.7))
        a, b = poses(hip, orc, pos.astype(np.float32), d.astype(np.float32), (0.0, 0.0, 1.0), h, v)
        si = hip.InputState(*keys, cur[0], cur[1], 1920, 1080)
        so = orc.InputState(*keys, cur[0], cur[1], 1920, 1080)
        delta = float(rng.uniform(1e-4, 0.1))
        hip.camera_handle_input(a, si, delta)
        L.orc_camera_handle_input(C.byref(b), C.byref(so), delta)
        assert same(a, b), (keys, cur, delta)
        # Camera::update (camera.cpp:46-52) of the new angles
        dir_p = hip.camera_update(a.horizontal_angle, a.vertical_angle)
        L.orc_camera_update(C.byref(b))
        assert np.asarray(dir_p, dtype=np.float32).tobytes() == bytes(b.direction)


def test_handle_input_cases(hip, orc):
    f32 = np.float32
    # W: position += direction * speed * float(delta); LEFT_SHIFT: speed 40; cursor at the centre: angles unchanged
    a, _ = poses(hip, orc, (1, 30, 90), (1, 0, 0), (0, 0, 1), 0.25, -0.1)
    hip.camera_handle_input(a, hip.InputState(1, 0, 0, 0, 0, 0, 1, 0, 960.0, 540.0, 1920, 1080), 0.016)
    assert a.position[0] == f32(1) + f32(1) * f32(40) * f32(0.016) and a.position[1] == 30 and a.position[2] == 90
    assert a.horizontal_angle == 0.25 and a.vertical_angle == -0.1
    # W wins over S, A over D, SPACE over LEFT_CONTROL (the else-ifs of camera.cpp:9-26)
    a, _ = poses(hip, orc, (0, 0, 0), (0, 1, 0), (0, 0, 1), 0.0, 0.0)
    hip.camera_handle_input(a, hip.InputState(1, 1, 1, 1, 1, 1, 0, 1, 0.0, 0.0, 1920, 1080), 0.5)
    right = np.cross(np.array([0, 1, 0], dtype=np.float32), np.array([0, 0, 1], dtype=np.float32))  # (1, 0, 0)
    assert tuple(a.position) == (-float(right[0]) * 0.5, 0.5, 0.5)
    assert a.horizontal_angle == 0.0 and a.vertical_angle == 0.0  # LEFT_ALT: no mouse look (camera.cpp:27-29)
    # mouse look: 0.012 per pixel from the centre; the vertical angle is clamped to +-(pi/2 - 0.001) with pi a float
    a, _ = poses(hip, orc, (0, 0, 0), (0, 1, 0), (0, 0, 1), 1.0, 0.0)
    hip.camera_handle_input(a, hip.InputState(0, 0, 0, 0, 0, 0, 0, 0, 960.0 + 100.0, 540.0 - 1000.0, 1920, 1080), 0.016)
    assert a.horizontal_angle == 1.0 + 100.0 * 0.012
    assert a.vertical_angle == float(f32(np.pi) / f32(2)) - 0.001
    assert tuple(a.position) == (0.0, 0.0, 0.0)
    L = hip.lib()
    assert L.tyr_camera_handle_input(None, None, 0.0) == -1


@pytest.mark.gpu
def test_flythrough_writes_the_references_performance_file(hip, tmp_path):
    """examples/flythrough.cpp: three recorded views (performance_measure.h:4-5), launch_kernels + swap per frame, the
    reference's Performance.txt layout; between the views the camera walks under scripted input, which resets the
    accumulation every frame (kernel.cu:702-718), while a parked camera accumulates"""
    exe = os.path.join(ROOT, "tyrant_amd", "bin", "flythrough")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tyrant_amd", "csrc"), "example"], check=True)
    out = str(tmp_path / "Performance.txt")
    p = subprocess.run([exe, "0", "1.0", out], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = open(out).read().split("\n")
    assert len(lines) == 15 and lines[4] == "" and lines[9] == "" and lines[14] == ""  # 3 x 4 lines, a blank line between views
    for v in range(3):
        blk = lines[5 * v : 5 * v + 4]
        assert [b.split(":")[0] for b in blk] == ["Average ms", "Average fps", "Min ms", "Max ms"]
        avg_ms, fps, mn, mx = (float(b.split(":")[1]) for b in blk)
        assert avg_ms > 0 and abs(fps - 1000.0 / avg_ms) / fps < 1e-3
        assert 0 < mn <= mx < 1.0  # the reference prints these two in seconds (performance_measure.cpp:30-31)
    views = [l for l in p.stdout.splitlines() if l.startswith("view ")]
    walks = [l for l in p.stdout.splitlines() if l.startswith("walked ")]
    assert len(views) == 3 and len(walks) == 2
    for l in views:  # a parked camera accumulates: many finished paths in the probe pixel
        assert float(l.split("accumulated ")[1].split(" ")[0]) > 8
    for l in walks:  # a moving camera resets every frame: at most what ONE launch_kernels finishes for that pixel
        assert float(l.split("holds ")[1].split(" ")[0]) <= 2
    assert "device_error 0" in p.stdout


@pytest.mark.gpu
def test_flythrough_progressive_display_writes_pictures_and_the_hud(hip, tmp_path):
    """the interactive build's progressive display (main.cpp:164-203), headless: every K-th frame the tone-mapped picture of the
    accumulation so far (kernel.cu:648-662) as a PPM, and the "Performance" window's text (main.cpp:177-198): the frame-rate
    line, the last <= 200 frame times, camera and sun"""
    exe = os.path.join(ROOT, "tyrant_amd", "bin", "flythrough")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tyrant_amd", "csrc"), "example"], check=True)
    prefix = str(tmp_path / "shot")
    p = subprocess.run([exe, "0", "0.3", str(tmp_path / "Performance.txt"), "150", prefix], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    shots = sorted(f for f in os.listdir(tmp_path) if f.startswith("shot_") and f.endswith(".ppm"))
    assert len(shots) >= 2 and all(int(f[5:-4]) % 150 == 0 for f in shots)
    data = open(tmp_path / shots[-1], "rb").read()
    assert data.startswith(b"P6") and len(data) > 1920 * 1080 * 3
    import numpy as np

    px = np.frombuffer(data[-1920 * 1080 * 3:], dtype=np.uint8)
    assert px.max() > 32 and px.std() > 4  # a picture, not a constant frame
    hud = open(prefix + "_hud.txt").read().split("\n")
    assert hud[0].startswith("Application average ") and "ms/frame" in hud[0] and "FPS" in hud[0]
    n = int(hud[1].split("(")[1].rstrip(")"))
    assert hud[1].startswith("Frametimes") and 1 <= n <= 200
    times = [float(x) for x in hud[2 : 2 + n]]
    assert all(0 < t < 1.0 for t in times)
    assert hud[2 + n].startswith("X: ") and hud[3 + n].startswith("Hor: ") and hud[4 + n].startswith("Sun X: ")
    assert "progressive display:" in p.stdout and "device_error 0" in p.stdout
