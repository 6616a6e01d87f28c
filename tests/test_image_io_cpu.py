"""pt/image_io.hpp — the C++ host's own PNG / baseline-JPEG / PPM decoders and its PNG writer (a main.cpp-style
C++ caller opens the reference's own images/Xilinx.jpg and images/SYCL.png and writes out.png, without stb).  Every decoded pixel is
compared with PIL's — the loader of the Python host, so both hosts put the same texels into the atlas — on the reference's two images
(when /root/reference is present: the build container) and on generated files: chroma sub-sampling 4:4:4 / 4:2:2 / 4:2:0 / 4:4:0, restart
intervals, sizes that are not multiples of the MCU, grey JPEG; PNG colour types grey / RGB / palette / with alpha, 8 and 16 bits, all five
filter types, multi-chunk IDAT.  Unsupported files fail with a reason (the caller then falls back like texture.hpp:106-111)."""
import io
import subprocess
import zlib
import struct
from pathlib import Path

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")
ROOT = Path(__file__).resolve().parent.parent
REF_IMAGES = Path("/root/reference/images")


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    out = tmp_path_factory.mktemp("imgio") / "image_io_main"
    subprocess.run(["g++", "-std=c++20", "-O2", "-Wall", "-Wextra", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    f"-I{ROOT / 'path_tracer_amd' / 'include'}", str(ROOT / "tests" / "cpp" / "image_io_main.cpp"), "-o", str(out)], check=True)
    return out


def decode(tool, path, tmp_path):
    out = tmp_path / "out.rgb"
    r = subprocess.run([str(tool), "decode", str(path), str(out)], capture_output=True, text=True)
    if r.returncode == 3:
        return r.stderr.strip()
    assert r.returncode == 0, r.stderr
    raw = out.read_bytes()
    head, body = raw.split(b"\n", 1)
    w, h = map(int, head.split())
    return np.frombuffer(body, np.uint8).reshape(h, w, 3)


def pil_rgb(path):
    return np.asarray(PIL.open(path).convert("RGB"))


def picture(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(x / 7.0 + seed), 128 + 90 * np.cos(y / 5.0), (x * 3 + y * 5) % 256], -1)
    img = base + rng.normal(0, 12, (h, w, 3))
    img[h // 3:h // 3 + 5, :, :] = [255, 0, 0]          # hard edges: chroma up-sampling and clamping get exercised
    img[:, w // 2:w // 2 + 3, :] = [0, 255, 255]
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.skipif(not REF_IMAGES.exists(), reason="the reference tree is only present in the build container")
def test_the_references_own_images_decode_like_the_python_host(tool, tmp_path):
    for name, shape in (("Xilinx.jpg", (512, 1024, 3)), ("SYCL.png", (559, 1280, 3))):
        got = decode(tool, REF_IMAGES / name, tmp_path)
        assert not isinstance(got, str), got
        assert got.shape == shape
        np.testing.assert_array_equal(got, pil_rgb(REF_IMAGES / name))
    # ... and they are the texels of the committed fixture both hosts' tests build config 1 from
    z = np.load(ROOT / "tests" / "golden" / "cfg1_textures.npz")
    keys = sorted(z.files)
    arrays = {k: z[k] for k in keys}
    assert any(a.shape == (512, 1024, 3) and np.array_equal(a, pil_rgb(REF_IMAGES / "Xilinx.jpg")) for a in arrays.values()), keys


@pytest.mark.parametrize("sub,size,restart,quality", [(0, (64, 48), 0, 90), (0, (37, 23), 0, 75), (1, (64, 48), 0, 85), (1, (37, 23), 3, 60),
                                                      (2, (64, 48), 0, 95), (2, (37, 23), 0, 75), (2, (130, 71), 5, 40), (2, (1, 1), 0, 75),
                                                      (2, (17, 2), 0, 75), (1, (2, 19), 0, 75), (0, (8, 8), 1, 100)])
def test_jpeg_pixels_equal_libjpegs(tool, tmp_path, sub, size, restart, quality):
    img = picture(size[0], size[1], sub * 10 + size[0])
    p = tmp_path / "t.jpg"
    kw = dict(quality=quality, subsampling=sub)
    if restart:
        kw["restart_marker_blocks"] = restart
    PIL.fromarray(img).save(p, "JPEG", **kw)
    got = decode(tool, p, tmp_path)
    assert not isinstance(got, str), got
    np.testing.assert_array_equal(got, pil_rgb(p))


def test_grey_optimised_and_unsupported_jpegs(tool, tmp_path):
    img = picture(53, 31, 4)
    PIL.fromarray(img).convert("L").save(tmp_path / "g.jpg", "JPEG", quality=80)
    np.testing.assert_array_equal(decode(tool, tmp_path / "g.jpg", tmp_path), pil_rgb(tmp_path / "g.jpg"))
    PIL.fromarray(img).save(tmp_path / "o.jpg", "JPEG", quality=70, optimize=True)  # its own Huffman tables
    np.testing.assert_array_equal(decode(tool, tmp_path / "o.jpg", tmp_path), pil_rgb(tmp_path / "o.jpg"))
    PIL.fromarray(img).save(tmp_path / "p.jpg", "JPEG", progressive=True)
    assert "progressive" in decode(tool, tmp_path / "p.jpg", tmp_path)
    PIL.fromarray(img).convert("CMYK").save(tmp_path / "c.jpg", "JPEG")
    assert isinstance(decode(tool, tmp_path / "c.jpg", tmp_path), str)
    (tmp_path / "t.jpg").write_bytes((tmp_path / "o.jpg").read_bytes()[:300])
    assert isinstance(decode(tool, tmp_path / "t.jpg", tmp_path), str)  # truncated: a reason, not a crash (the tool runs under ASan)
    assert decode(tool, tmp_path / "missing.jpg", tmp_path) == "can't fopen"
    (tmp_path / "x.bin").write_bytes(b"GIF89a....")
    assert "unknown image type" in decode(tool, tmp_path / "x.bin", tmp_path)


def raw_png(w, h, ctype, depth, rows, filters, palette=None, idat_chunks=1):
    """A PNG assembled by hand so that every filter type is used: rows = unfiltered scan-lines (bytes)."""
    bpp = max(1, {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype] * depth // 8)
    prev = bytes(len(rows[0]))
    out = b""
    for r, ft in zip(rows, filters):
        line = bytearray(len(r))
        for i in range(len(r)):
            a = r[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
            pred = [0, a, b, (a + b) // 2, a if pa <= pb and pa <= pc else (b if pb <= pc else c)][ft]
            line[i] = (r[i] - pred) & 255
        out += bytes([ft]) + bytes(line)
        prev = r
    z = zlib.compress(out, 9)
    chunk = lambda t, d: struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)  # noqa: E731
    body = chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if palette is not None:
        body += chunk(b"PLTE", bytes(palette))
    step = (len(z) + idat_chunks - 1) // idat_chunks
    for k in range(0, len(z), step):
        body += chunk(b"IDAT", z[k:k + step])
    return b"\x89PNG\r\n\x1a\n" + body + chunk(b"IEND", b"")


def test_png_colour_types_depths_and_filters(tool, tmp_path):
    img = picture(45, 29, 9)
    h, w, _ = img.shape
    filters = [k % 5 for k in range(h)]
    cases = {
        "rgb8": (2, 8, [img[y].tobytes() for y in range(h)]),
        "rgba8": (6, 8, [np.concatenate([img[y], np.full((w, 1), 200, np.uint8)], 1).tobytes() for y in range(h)]),
        "grey8": (0, 8, [img[y, :, 0].tobytes() for y in range(h)]),
        "greya8": (4, 8, [np.stack([img[y, :, 1], img[y, :, 2]], 1).tobytes() for y in range(h)]),
        "rgb16": (2, 16, [np.stack([img[y], img[y][:, ::-1]], -1).tobytes() for y in range(h)]),  # high byte = the picture
        "grey16": (0, 16, [np.stack([img[y, :, 0], img[y, :, 1]], -1).tobytes() for y in range(h)]),
    }
    for name, (ctype, depth, rows) in cases.items():
        p = tmp_path / f"{name}.png"
        p.write_bytes(raw_png(w, h, ctype, depth, rows, filters, idat_chunks=3))
        got = decode(tool, p, tmp_path)
        assert not isinstance(got, str), (name, got)
        want = pil_rgb(p) if depth == 8 else None
        if name == "rgb16":
            want = img
        if name == "grey16":
            want = np.repeat(img[:, :, :1], 3, 2)
        np.testing.assert_array_equal(got, want, err_msg=name)
    # palette, 8 / 4 / 2 / 1 bits, and sub-byte grey
    pal = np.random.default_rng(3).integers(0, 256, (16, 3), dtype=np.uint8)
    for depth in (8, 4, 2, 1):
        idx = (img[:, :, 0].astype(np.int32) * ((1 << min(depth, 4)) - 1) // 255).astype(np.uint8)
        per = 8 // depth
        rows = []
        for y in range(h):
            bits = np.zeros((w + per - 1) // per * per, np.uint8); bits[:w] = idx[y]
            packed = np.zeros(len(bits) // per, np.uint8)
            for k in range(per):
                packed |= bits[k::per] << ((per - 1 - k) * depth)
            rows.append(packed.tobytes())
        p = tmp_path / f"pal{depth}.png"
        p.write_bytes(raw_png(w, h, 3, depth, rows, filters, palette=pal.reshape(-1)))
        np.testing.assert_array_equal(decode(tool, p, tmp_path), pil_rgb(p), err_msg=f"palette {depth}")
        if depth < 8:
            p = tmp_path / f"grey{depth}.png"
            p.write_bytes(raw_png(w, h, 0, depth, rows, filters))
            want = np.repeat((idx.astype(np.int32) * (255 // ((1 << depth) - 1))).astype(np.uint8)[:, :, None], 3, 2)
            np.testing.assert_array_equal(decode(tool, p, tmp_path), want, err_msg=f"grey {depth}")
    # PIL-written files (adaptive filters, real compression), a corrupt CRC, an interlaced file
    PIL.fromarray(img).save(tmp_path / "pil.png", optimize=True)
    np.testing.assert_array_equal(decode(tool, tmp_path / "pil.png", tmp_path), img)
    bad = bytearray((tmp_path / "pil.png").read_bytes()); bad[60] ^= 1
    (tmp_path / "bad.png").write_bytes(bytes(bad))
    assert isinstance(decode(tool, tmp_path / "bad.png", tmp_path), str)
    inter = bytearray(raw_png(w, h, 2, 8, cases["rgb8"][2], filters)); inter[28] = 1
    inter[29:33] = struct.pack(">I", zlib.crc32(bytes(inter[12:29])) & 0xffffffff)
    (tmp_path / "i.png").write_bytes(bytes(inter))
    assert "interlaced" in decode(tool, tmp_path / "i.png", tmp_path)


@pytest.mark.parametrize("size", [(1, 1), (45, 29), (400, 225)])
def test_png_writer_round_trips(tool, tmp_path, size):
    img = picture(size[0], size[1], 11)
    (tmp_path / "in.rgb").write_bytes(img.tobytes())
    out = tmp_path / "out.png"
    subprocess.run([str(tool), "encode", str(size[0]), str(size[1]), str(tmp_path / "in.rgb"), str(out)], check=True)
    np.testing.assert_array_equal(pil_rgb(out), img)                              # any PNG reader decodes the exact bytes
    np.testing.assert_array_equal(decode(tool, out, tmp_path), img)               # ... including this host's own
    from path_tracer_amd import png
    png.write_png(str(tmp_path / "py.png"), img)
    np.testing.assert_array_equal(decode(tool, tmp_path / "py.png", tmp_path), img)  # and the Python host's writer is read back


def test_corrupt_and_hostile_files_fail_with_a_reason(tool, tmp_path):
    """truncated, bit-flipped and size-lying files must come back as a failure reason — no crash, no sanitizer report
    (the tool is built with -fsanitize=address,undefined), no allocation sized by a hostile header."""
    rng = np.random.default_rng(11)
    files = []
    seeds = []
    for k, (fmt, kw) in enumerate([("JPEG", dict(quality=80, subsampling=2)), ("JPEG", dict(quality=60, subsampling=0, restart_marker_blocks=2)),
                                    ("PNG", {}), ("PNG", dict(compress_level=0))]):
        p = tmp_path / f"seed{k}.{fmt.lower()}"
        PIL.fromarray(picture(41, 29, k)).save(p, fmt, **kw)
        seeds.append(p.read_bytes())
    n = 0
    for k, data in enumerate(seeds):
        for cut in sorted(set(int(c) for c in np.linspace(1, len(data) - 1, 24))):   # truncations
            q = tmp_path / f"t{k}_{cut}"; q.write_bytes(data[:cut]); files.append(q)
        for _ in range(60):                                                            # bit flips (1 - 4 per file)
            b = bytearray(data)
            for _ in range(int(rng.integers(1, 5))):
                b[int(rng.integers(2, len(b)))] ^= 1 << int(rng.integers(0, 8))
            q = tmp_path / f"f{k}_{n}"; q.write_bytes(bytes(b)); files.append(q); n += 1
    # a 65535 x 65535 JPEG frame header on a tiny file: rejected before anything is allocated
    jpg = bytearray(seeds[0])
    sof = jpg.index(b"\xff\xc0")
    jpg[sof + 5:sof + 9] = b"\xff\xff\xff\xff"
    q = tmp_path / "huge.jpg"; q.write_bytes(bytes(jpg)); files.append(q)
    # 16-bit quantisation tables of 0xffff + the largest DC differences: the products stay in range (UBSan would report the overflow)
    jpg = bytearray(seeds[1])
    dqt = jpg.index(b"\xff\xdb")
    ln = (jpg[dqt + 2] << 8) | jpg[dqt + 3]
    table = bytes([0x10]) + b"\xff\xff" * 64                   # Pq = 1 (16 bit), Tq = 0
    jpg[dqt:dqt + 2 + ln] = b"\xff\xdb" + struct.pack(">H", 2 + len(table)) + table
    q = tmp_path / "q16.jpg"; q.write_bytes(bytes(jpg)); files.append(q)
    # a PNG whose IDAT expands far beyond (stride + 1) * height: stopped at the image's own size
    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)
    bomb = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * (64 << 20), 9)) + chunk(b"IEND", b"")
    q = tmp_path / "bomb.png"; q.write_bytes(bomb); files.append(q)
    r = subprocess.run([str(tool), "many"] + [str(f) for f in files], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(files)
    by = dict(zip([f.name for f in files], lines))
    assert by["huge.jpg"] == "err JPEG too large", by["huge.jpg"]
    assert by["bomb.png"].startswith("err deflate stream larger"), by["bomb.png"]
    assert by["q16.jpg"].startswith(("ok", "err")), by["q16.jpg"]
    assert all(l.startswith(("ok ", "err ")) for l in lines)
    assert sum(l.startswith("err") for l in lines) > len(lines) // 4      # truncations and most flips in headers / CRCs are refused
