"""The RIFF/WAVE reader of the drop-in surface (`evaluate.load_audio`, used by `Translator.predict(path)` and the
`m4t_evaluate` data path).  The reference decodes files with fairseq2's AudioDecoder = libsndfile
(inference/translator.py:135, 270-273; cli/m4t/evaluate/evaluate.py:157), which is not in this image: what is pinned here is
libsndfile's documented float conversion - integer samples scaled by 2^-(bits-1), G.711 codes expanded to 16-bit linear
first (checked against the standard library's `audioop`) - on every encoding the reader accepts."""
import audioop
import struct
import wave

import numpy as np
import pytest

from seamless_communication_amd.evaluate import load_audio


def _chunks(fmt_body: bytes, data: bytes) -> bytes:
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt_body)) + fmt_body + b"data" + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b"")
    return b"RIFF" + struct.pack("<I", len(body)) + body


def _fmt(tag, channels, rate, bits, extensible=False):
    width = bits // 8
    base = struct.pack("<HHIIHH", 0xFFFE if extensible else tag, channels, rate, rate * channels * width, channels * width, bits)
    if not extensible:
        return base
    guid = struct.pack("<H", tag) + bytes.fromhex("000000001000800000aa00389b71")
    return base + struct.pack("<HHI", 22, bits, 0) + guid


@pytest.mark.parametrize("width", [1, 2, 3, 4])
def test_integer_pcm_written_by_the_wave_module(tmp_path, width):
    rng = np.random.default_rng(width)
    n, ch = 1000, 2
    full = 1 << (8 * width - 1)
    ints = rng.integers(-full, full, size=(n, ch), dtype=np.int64)
    ints[0] = [-full, full - 1]
    if width == 1:
        raw = (ints + 128).astype(np.uint8).tobytes()
    else:
        raw = b"".join(int(v).to_bytes(width, "little", signed=True) for v in ints.reshape(-1))
    p = tmp_path / f"pcm{8 * width}.wav"
    with wave.open(str(p), "wb") as w:
        w.setnchannels(ch), w.setsampwidth(width), w.setframerate(22050)
        w.writeframes(raw)
    x, rate = load_audio(p, all_channels=True)
    assert rate == 22050 and x.shape == (n, ch) and x.dtype == np.float32
    want = (ints.astype(np.float64) / full).astype(np.float32)
    assert np.array_equal(x, want)
    mono, _ = load_audio(p)
    assert np.array_equal(mono, want[:, 0])
    assert x.min() == -1.0 and x.max() <= 1.0  # (2^31 - 1) / 2^31 rounds to 1.0 in float32, as in libsndfile


@pytest.mark.parametrize("bits,dtype", [(32, "<f4"), (64, "<f8")])
@pytest.mark.parametrize("extensible", [False, True])
def test_ieee_float(tmp_path, bits, dtype, extensible):
    x = np.random.default_rng(bits).standard_normal((333, 3)).astype(dtype) * 0.3
    p = tmp_path / "f.wav"
    p.write_bytes(_chunks(_fmt(3, 3, 16000, bits, extensible), x.tobytes()))
    y, rate = load_audio(p, all_channels=True)
    assert rate == 16000 and np.array_equal(y, x.astype(np.float32))


@pytest.mark.parametrize("tag,expand", [(7, audioop.ulaw2lin), (6, audioop.alaw2lin)])
def test_g711_codes_expand_like_audioop(tmp_path, tag, expand):
    codes = bytes(range(256)) * 2
    p = tmp_path / "g711.wav"
    p.write_bytes(_chunks(_fmt(tag, 1, 8000, 8), codes))
    y, rate = load_audio(p)
    lin = np.frombuffer(expand(codes, 2), dtype="<i2").astype(np.float32) / 32768.0
    assert rate == 8000 and np.array_equal(y, lin)


def test_extensible_pcm16_and_odd_sized_chunks(tmp_path):
    x = np.arange(-5, 6, dtype="<i2")  # 11 samples of one channel: 22 bytes
    junk = b"LIST" + struct.pack("<I", 3) + b"abc\0"  # an odd-sized chunk in front of fmt: padded to even
    body = b"WAVE" + junk + b"fmt " + struct.pack("<I", 40) + _fmt(1, 1, 16000, 16, True) + b"data" + struct.pack("<I", 22) + x.tobytes()
    p = tmp_path / "e.wav"
    p.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)
    y, rate = load_audio(p)
    assert rate == 16000 and np.array_equal(y, x.astype(np.float32) / 32768.0)


def test_errors_name_the_problem(tmp_path):
    p = tmp_path / "a.flac"
    p.write_bytes(b"fLaC" + b"\0" * 64)
    with pytest.raises(ValueError, match="FLAC audio needs the `soundfile` package"):  # not installed in this image
        load_audio(p)
    p = tmp_path / "b.wav"
    p.write_bytes(_chunks(_fmt(2, 1, 16000, 4), b"\0" * 16))  # ADPCM
    with pytest.raises(ValueError, match="unsupported WAVE encoding"):
        load_audio(p)
    p = tmp_path / "c.wav"
    p.write_bytes(b"not a wave file at all")
    with pytest.raises(ValueError, match="not RIFF/WAVE"):
        load_audio(p)


def test_other_containers_go_through_soundfile_when_it_is_importable(tmp_path, monkeypatch):
    """FLAC / Ogg / AIFF ...: the reference's AudioDecoder is libsndfile; `load_audio` hands such files to the `soundfile` package when
    the environment has it (here: a stand-in module, the real one is not installed in this image) and keeps the file's own rate."""
    import sys
    import types

    import numpy as np

    calls = []

    def fake_read(path, dtype="float64", always_2d=False):
        calls.append((path, dtype, always_2d))
        return np.stack([np.linspace(-0.5, 0.5, 100, dtype=np.float32), np.zeros(100, dtype=np.float32)], axis=1), 22050

    monkeypatch.setitem(sys.modules, "soundfile", types.SimpleNamespace(read=fake_read))
    p = tmp_path / "a.flac"
    p.write_bytes(b"fLaC" + b"\0" * 64)
    mono, rate = load_audio(p)
    assert rate == 22050 and mono.shape == (100,) and mono.dtype == np.float32 and mono[0] == -0.5
    both, _ = load_audio(p, all_channels=True)
    assert both.shape == (100, 2) and calls[-1] == (str(p), "float32", True)
