"""analyze() end to end on a GPU box: WAV in (16 kHz mono and 48 kHz stereo), reference-format CSV out,
resume after an interrupted run, manifest lock."""
import os
import wave

import numpy as np
import pytest

from oracle import yamnet_oracle as O
from oracle import resample_oracle as RO


def write_wav(path, x, rate):
    x = np.asarray(x)
    if x.ndim == 1:
        x = x[:, None]
    with wave.open(str(path), "wb") as w:
        w.setnchannels(x.shape[1])
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes((np.clip(x, -1, 1 - 2 ** -15) * 32768.0).round().astype("<i2").tobytes())


def test_wav_reader_roundtrip(tmp_path):
    from buzzdetect_amd.analyze import WavTrack, build_ident
    x = (np.arange(2000).reshape(1000, 2) % 97 - 48) / 64.0
    write_wav(tmp_path / "a.wav", x, 32000)
    t = WavTrack(str(tmp_path / "a.wav"))
    assert (t.samplerate, t.channels, t.frames) == (32000, 2, 1000) and t.duration == 1000 / 32000
    t.seek(10)
    got = t.read(5)
    assert got.shape == (5, 2) and np.allclose(got, x[10:15], atol=1 / 32768)
    t.seek(998)
    assert t.read(10).shape == (2, 2)              # short read at the end of the file
    t.close()
    assert build_ident("/data/audio/site1/rec.wav", "/data/audio") == "site1/rec"


@pytest.mark.gpu
def test_analyze_writes_reference_format_and_resumes(engine, weights_bundle, tmp_path):
    import pandas as pd
    from buzzdetect_amd.analyze import analyze
    from buzzdetect_amd import results as R
    audio, out = tmp_path / "audio", tmp_path / "out"
    (audio / "site").mkdir(parents=True)
    x16 = O.synthetic_audio(16000 * 9 + 123, seed=8)
    write_wav(audio / "site" / "mono16.wav", x16, 16000)
    t = np.arange(48000 * 5) / 48000.0
    st = np.stack([0.3 * np.sin(2 * np.pi * 300 * t), 0.2 * np.sin(2 * np.pi * 1500 * t)], 1)
    write_wav(audio / "stereo48.wav", st, 48000)

    rep = analyze("model_general_v3", classes_out=["ins_buzz", "ambient_rain"], framehop_prop=1, chunklength=3,
                  dir_audio=str(audio), dir_out=str(out), engine=engine)
    assert rep.files_done == 2 and rep.files_total == 2
    a = pd.read_csv(out / "site" / "mono16_buzzdetect.csv")
    assert list(a.columns) == ["start", "activation_ambient_rain", "activation_ins_buzz"]   # model order (H5)
    assert not (out / "site" / "mono16_buzzpart.csv").exists()
    # chunklength 3 -> 2.88 s = 3 frames; 9.0077 s -> chunks [0,2.88) [2.88,5.76) [5.76,8.64) [8.64,9.01)
    assert a["start"].tolist() == [0.0, 0.96, 1.92, 2.88, 3.84, 4.8, 5.76, 6.72, 7.68, 8.64]
    # values: the 16-bit samples the reader produced, chunked the same way, through the f64 oracle
    b = weights_bundle
    x_q = (np.clip(x16, -1, 1 - 2 ** -15) * 32768.0).round().astype(np.int16).astype(np.float32) / 32768.0
    ref = O.predict(x_q[: int(2.88 * 16000)], b["blob"], b["mel"], b["head_kernel"], b["head_bias"], 15360, 96, np.float64)
    assert np.abs(a["activation_ins_buzz"].to_numpy()[:3] - ref[:, 8].round(2)).max() <= 0.011
    s = pd.read_csv(out / "stereo48_buzzdetect.csv")
    assert s["start"].tolist() == [0.0, 0.96, 1.92, 2.88, 3.84, 4.8]        # chunks [0,2.88) + [2.88,5.0): 3 + 3 rows
    q = (np.clip(st, -1, 1 - 2 ** -15) * 32768.0).round().astype(np.int16).astype(np.float32) / 32768.0
    mono = RO.resample(q[: int(2.88 * 48000)], 48000).astype(np.float32)
    ref48 = O.predict(mono, b["blob"], b["mel"], b["head_kernel"], b["head_bias"], 15360, 96, np.float64)
    assert np.abs(s["activation_ins_buzz"].to_numpy()[:3] - ref48[:, 8].round(2)).max() <= 0.011

    # second run: nothing to do; a different setting in the same folder is refused by the manifest
    rep2 = analyze("model_general_v3", classes_out=["ins_buzz", "ambient_rain"], chunklength=3,
                   dir_audio=str(audio), dir_out=str(out), engine=engine)
    assert rep2.files_done == 0 and rep2.files_skipped == 2
    with pytest.raises(RuntimeError, match="different settings"):
        analyze("model_general_v3", precision=0.95, chunklength=3, dir_audio=str(audio), dir_out=str(out), engine=engine)

    # interrupted run: keep only the rows of chunks 0 and 2 as a partial file -> only the gaps are redone
    os.remove(out / "site" / "mono16_buzzdetect.csv")
    a.iloc[[0, 1, 2, 6, 7, 8]].to_csv(out / "site" / "mono16_buzzpart.csv", index=False)
    rep3 = analyze("model_general_v3", classes_out=["ins_buzz", "ambient_rain"], chunklength=3,
                   dir_audio=str(audio), dir_out=str(out), engine=engine)
    # gaps are (2.88, 5.76) and (8.64, 9.0077); like the reference (results_coverage.py:45-46) a gap that
    # starts within one frame of the end is ignored, so one chunk is redone and the 8.64 row stays missing
    assert rep3.files_done == 1 and rep3.chunks == 1
    a3 = pd.read_csv(out / "site" / "mono16_buzzdetect.csv")
    assert a3.equals(a.iloc[:9].reset_index(drop=True))

    # detections mode in a fresh folder
    out2 = tmp_path / "out_det"
    analyze("model_general_v3", precision=0.95, chunklength=200, dir_audio=str(audio), dir_out=str(out2), engine=engine)
    d = pd.read_csv(out2 / "site" / "mono16_buzzdetect.csv")
    assert list(d.columns) == ["start", "detections_ins_buzz"] and set(d["detections_ins_buzz"]) <= {0, 1}
    assert R.threshold_for_precision("model_general_v3", 0.95) == pytest.approx(-1.205)
