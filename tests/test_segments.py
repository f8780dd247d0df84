"""CPU: the data formats / caller loop either side of the path (fish_diffusion_amd/segments.py): .npy feature dicts, silence
slicing (utils/audio.py:112-167), the paste of inference.py:336-376, PCM_16 .wav output."""
import math
import wave

import numpy as np
import torch

from fish_diffusion_amd import segments as S

SR = 44100


def _recording():
    t = np.arange(SR * 5) / SR
    a = np.zeros_like(t, dtype=np.float32)
    a[SR:2 * SR] = 0.5 * np.sin(2 * np.pi * 220 * t[SR:2 * SR])
    a[int(3.2 * SR):int(4.5 * SR)] = 0.3 * np.sin(2 * np.pi * 330 * t[int(3.2 * SR):int(4.5 * SR)])
    a[int(2.5 * SR):int(2.5 * SR) + 2000] = 0.4          # a 45 ms click: "too short, unlikely to be vocal"
    return a


def test_split_and_slice_follow_the_reference_logic():
    a = _recording()
    iv = S.split_nonsilent(a, top_db=60)
    assert iv.shape == (3, 2) and (iv % 512 == 0).all()                      # frame-aligned edges (frames_to_samples)
    for (s, e), (lo, hi) in zip(iv, ((SR, 2 * SR), (int(2.5 * SR), int(2.5 * SR) + 2000), (int(3.2 * SR), int(4.5 * SR)))):
        assert abs(s - lo) <= 2048 and abs(e - hi) <= 2048                      # within one analysis frame of the true onsets
    segs = list(S.slice_audio(a, SR, max_duration=30.0))
    assert segs == [tuple(iv[0]), tuple(iv[2])]                                # the click (< 0.1 s) is dropped, audio.py:155-157
    # long intervals are cut into ceil(len / max) equal chunks; the last one may run past the interval's end (audio.py:162-167)
    segs = list(S.slice_audio(a, SR, max_duration=0.5))
    for s0, e0 in (iv[0], iv[2]):
        n = math.ceil((e0 - s0) / (0.5 * SR))
        size = math.ceil((e0 - s0) / n)
        mine = [sg for sg in segs if s0 <= sg[0] < e0]
        assert mine == [(i, i + size) for i in range(s0, e0, size)]
    # merging across short silences (audio.py:138-151)
    merged = list(S.slice_audio(a, SR, max_duration=30.0, min_silence_duration=1.5))
    assert merged == [(int(iv[0][0]), int(iv[2][1]))]
    # all-silent and full-scale inputs
    assert list(S.slice_audio(np.zeros(SR, np.float32), SR)) == [(0, SR)]      # max(mse) = 0: every frame is at 0 dB re. the max
    assert list(S.slice_audio(np.ones(SR, np.float32), SR)) == [(0, SR)]


def test_stitch_is_the_reference_paste():
    total = 1000
    pieces = [(100, torch.arange(50.0)), (120, torch.ones(30) * 7), (980, torch.ones(64) * 3), (1000, torch.ones(5))]
    ref = np.zeros(total, np.float32)
    for start, wav in pieces:                                                   # inference.py:375-376, verbatim semantics
        w = wav.numpy()
        max_wav_len = total - start
        ref[start:start + w.shape[-1]] = w[:max_wav_len] if start + w.shape[-1] > total else w
    out = S.stitch(total, pieces)
    np.testing.assert_array_equal(out.numpy(), ref)
    assert S.stitch(10, []).shape == (10,)


def test_wav_and_sample_files_round_trip(tmp_path):
    a = np.sin(np.arange(5000) * 0.01).astype(np.float32) * 0.9
    a[10], a[11] = 1.5, -1.5                                                    # out of range: clipped
    p = tmp_path / "sub" / "out.wav"
    S.write_wav(str(p), torch.from_numpy(a), SR)                                # creates the directory, inference.py:384-385
    with wave.open(str(p), "rb") as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (1, 2, SR, 5000)
        pcm = np.frombuffer(f.readframes(5000), dtype="<i2")
    np.testing.assert_array_equal(pcm, np.clip(np.rint(a * 32767.0), -32768, 32767).astype(np.int16))
    assert pcm[10] == 32767 and pcm[11] == -32768
    sample = {"path": "x.wav", "audio": a, "sampling_rate": SR, "time_stretch": 1.0, "mel": np.zeros((128, 9), np.float32),
              "contents": np.ones((256, 5), np.float32), "pitches": np.full(9, 220.0, np.float32), "key_shift": 0}
    q = tmp_path / "x.0.data.npy"
    S.save_sample(str(q), sample)                                               # np.save(save_path, sample), extract_features.py:172
    back = S.load_sample(str(q))
    assert set(back) == set(sample) and back["sampling_rate"] == SR
    np.testing.assert_array_equal(back["contents"], sample["contents"])
    c, f0, mel = S.sample_to_device(back, torch.device("cpu"))
    assert c.shape == (256, 5) and f0.shape == (9,) and mel.shape == (128, 9)
