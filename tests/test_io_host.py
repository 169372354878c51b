"""Host-side harness I/O of include/vo_hip.h (test/vo_run.cpp's contract): PNG decoding against Pillow, associate.txt
parsing incl. the reference's end-of-file quirk, trajectory formatting (Eigen's default stream format), time report.
No GPU: these entry points are plain host C++."""
import numpy as np
import pytest


def test_png_reader_matches_pillow(vo, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    rgb[5:20, 7:30] = (rgb[5:20, 7:30] // 32) * 32  # some flat texture so that all filter types get picked
    Image.fromarray(rgb).save(tmp_path / "rgb.png")
    assert np.array_equal(vo.read_png(tmp_path / "rgb.png"), rgb)
    assert np.array_equal(vo.read_png(tmp_path / "rgb.png", as_bgr=True), rgb[:, :, ::-1])  # cv::imread(path, 1)
    rgba = rng.integers(0, 256, (16, 9, 4), dtype=np.uint8)
    Image.fromarray(rgba).save(tmp_path / "rgba.png")
    assert np.array_equal(vo.read_png(tmp_path / "rgba.png"), rgba)
    depth = rng.integers(0, 65536, (48, 64), dtype=np.uint16)
    depth[10:30, 10:40] = 5000
    Image.fromarray(depth).save(tmp_path / "depth.png")
    got = vo.read_png(tmp_path / "depth.png")
    assert got.dtype == np.uint16 and np.array_equal(got, depth)     # cv::imread(path, -1)
    gray = rng.integers(0, 256, (21, 34), dtype=np.uint8)
    Image.fromarray(gray).save(tmp_path / "gray.png", optimize=True)
    assert np.array_equal(vo.read_png(tmp_path / "gray.png"), gray)
    pal = Image.fromarray(rgb).quantize(16)
    pal.save(tmp_path / "pal.png")
    assert np.array_equal(vo.read_png(tmp_path / "pal.png"), np.asarray(pal.convert("RGB")))
    with pytest.raises(vo.VoError):
        vo.read_png(tmp_path / "missing.png")
    (tmp_path / "bad.png").write_bytes(b"not a png at all, definitely not")
    with pytest.raises(vo.VoError):
        vo.read_png(tmp_path / "bad.png")


def test_associate_file(vo, tmp_path):
    d = tmp_path / "seq"
    d.mkdir()
    lines = [f"{1305031102.175304 + 0.03 * i:.6f} rgb/{i}.png {1305031102.160407 + 0.03 * i:.6f} depth/{i}.png" for i in range(5)]
    (d / "associate.txt").write_text("\n".join(lines) + "\n")
    ds = vo.Dataset(str(d) + "/", 3)
    assert len(ds) == 3 and ds[2] == ("1305031102.235304", str(d) + "/rgb/2.png", "1305031102.220407", str(d) + "/depth/2.png")
    ds.close()
    # data_num beyond the file: eof is only noticed BEFORE a read (vo_run.cpp:45-49), so a file ending in a newline
    # yields one extra record of empty strings -- reproduced
    ds = vo.Dataset(str(d) + "/", 100)
    assert len(ds) == 6 and ds[5] == ("", str(d) + "/", "", str(d) + "/")
    ds.close()
    (d / "associate.txt").write_text("\n".join(lines))
    ds = vo.Dataset(str(d) + "/", 100)
    assert len(ds) == 5
    ds.close()
    with pytest.raises(vo.VoError):
        vo.Dataset(str(tmp_path / "nowhere"), 5)


def _eigen_row(v):
    s = ["%g" % x for x in v]
    w = max(len(t) for t in s)
    return " ".join(t.rjust(w) for t in s)


def test_trajectory_format_and_time_report(vo, tmp_path):
    rng = np.random.default_rng(1)
    T = np.concatenate([rng.normal(0, 2, (6, 3)), rng.normal(0, 1, (6, 4))], axis=1)
    T[:, 3:] /= np.linalg.norm(T[:, 3:], axis=1, keepdims=True)
    T[0, :3] = [1.0, -0.5, 123456.789]     # exercises the %g switch-over and the column alignment
    T[1, :3] = [1e-7, 0.0, -3.25]
    ts = [f"1305031102.{175304 + i}" for i in range(6)]
    vo.write_trajectory(tmp_path / "traj.txt", ts, T)
    got = (tmp_path / "traj.txt").read_text().splitlines()
    want = [f"{ts[i]} {_eigen_row(T[i, :3])} {_eigen_row(T[i, 3:])}" for i in range(6)]
    assert got == want
    back = np.array([[float(x) for x in ln.split()[1:]] for ln in got])
    assert np.allclose(back, T, rtol=1e-5, atol=1e-12)
    times = rng.uniform(0.01, 0.05, 11)
    med, mean = vo.tracking_time_stats(times)
    assert med == np.sort(times)[11 // 2] and abs(mean - times.sum() / 11) < 1e-15
