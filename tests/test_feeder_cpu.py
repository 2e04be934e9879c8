"""Batching semantics of the training feeder against the reference's rules (datafeeder.py:139-220): epoch walk,
length bucketing, padding (+1 frame, multiple of r), in-RAM cache, CMUDict substitution, the prefetch thread and
the round-robin dealing over data-parallel ranks.  Feature extraction is stubbed (it needs the GPU)."""
import os

import numpy as np

from nspeech_amd import hparams as hparams_mod
from nspeech_amd.datasets.datafeeder import DataFeeder, prepare_batch, _round_up
from nspeech_amd.utils.text import cmudict, sequence_to_text


def _corpus(tmp_path, n):
    os.makedirs(tmp_path / "wavs", exist_ok=True)
    words = ["hello", "world", "speech", "street", "turn", "left", "right", "now"]
    with open(tmp_path / "metadata.csv", "w") as f:
        for i in range(n):
            text = " ".join(words[(i * 3 + j) % len(words)] for j in range(2 + i % 5))
            f.write("utt%03d|raw|%s\n" % (i, text))
    return str(tmp_path)


def _stubs(hp):
    calls = []

    def loader(path):
        i = int(os.path.basename(path)[3:6])
        return np.zeros(250 * (20 + (i * 7) % 31), np.float32)          # 20..50 frames

    def features(wav):
        calls.append(len(wav))
        T = 1 + len(wav) // 250
        return np.full((hp.num_freq, T), 0.5, np.float32), np.full((hp.num_mels, T), 0.25, np.float32)
    return loader, features, calls


def _hp(**kw):
    hp = hparams_mod.load("taco2")
    hp.batch_size, hp.batch_group_size, hp.outputs_per_step = 4, 3, 5
    for k, v in kw.items():
        setattr(hp, k, v)
    return hp


def test_padding_rule_and_lengths():
    rng = __import__("random").Random(0)
    batch = [(np.arange(2, 2 + L, dtype=np.int32), 0, np.ones((T, 80), np.float32), np.ones((T, 1025), np.float32))
             for L, T in ((5, 17), (9, 24), (3, 20))]
    inputs, lengths, speakers, mel, lin = prepare_batch(batch, 5, rng)
    assert inputs.shape == (3, 9) and mel.shape == (3, 25, 80) and lin.shape == (3, 25, 1025)   # round_up(24 + 1, 5)
    assert sorted(lengths.tolist()) == [3, 5, 9]
    for i in range(3):
        assert (inputs[i, lengths[i]:] == 0).all() and (inputs[i, :lengths[i]] != 0).all()
        T = int(mel[i, :, 0].sum())
        assert (mel[i, T:] == 0).all() and (lin[i, T:] == 0).all()
    assert _round_up(25, 5) == 25 and _round_up(26, 5) == 30
    batch2 = [(np.arange(2, 6, dtype=np.int32), 0, np.ones((20, 80), np.float32), np.ones((20, 1025), np.float32))]
    assert prepare_batch(batch2, 5, rng)[3].shape[1] == 25                                         # 20 + 1 -> 25


def test_epoch_walk_bucketing_and_cache(tmp_path):
    hp = _hp()
    root = _corpus(tmp_path, 24)
    loader, features, calls = _stubs(hp)
    f = DataFeeder(hp, ljspeech=root, seed=3, prefetch=False, features=features, loader=loader)
    group = [f.next_batch() for _ in range(3)]            # one group = 3 batches of 4
    assert len(calls) == 12                                # 12 utterances processed, each once
    lens = [sorted(int(m[:, :, 0].sum(axis=1)[i]) for i in range(4)) for _, _, m, _ in group]
    flat = sorted(x for b in lens for x in b)
    runs = sorted(lens, key=lambda b: b[0])
    assert [x for b in runs for x in b] == flat            # batches are contiguous runs of the sorted group
    [f.next_batch() for _ in range(3)]                     # second group: the other 12 utterances
    assert len(calls) == 24
    [f.next_batch() for _ in range(3)]                     # wrap-around: reshuffled, served from the cache
    assert len(calls) == 24 and len(f.cache) == 24


def test_prefetch_thread_matches_synchronous(tmp_path):
    hp = _hp()
    root = _corpus(tmp_path, 30)
    out = []
    for prefetch in (False, True):
        loader, features, _ = _stubs(hp)
        f = DataFeeder(hp, ljspeech=root, seed=5, prefetch=prefetch, features=features, loader=loader)
        out.append([f.next_batch() for _ in range(7)])
    for a, b in zip(*out):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_rank_dealing(tmp_path):
    hp = _hp()
    root = _corpus(tmp_path, 40)
    per_rank = []
    for rank in range(2):
        loader, features, _ = _stubs(hp)
        f = DataFeeder(hp, ljspeech=root, seed=9, rank=rank, world=2, prefetch=False, features=features, loader=loader)
        batches = [f.next_batch() for _ in range(3)]
        per_rank.append(sorted(int(t) for _, _, m, _ in batches for t in m[:, :, 0].sum(axis=1)))
    # the two ranks split one sorted group of 24 alternately: their sorted lengths interleave
    merged = sorted(per_rank[0] + per_rank[1])
    assert merged[0::2] == per_rank[0] and merged[1::2] == per_rank[1]


def test_ranks_pad_to_the_global_batch(tmp_path):
    """the loss is an unmasked mean, so the mean of the ranks' losses is the global loss only if the ranks
    pad to the same T_out.  Global step k is the same slice of the sorted group on every rank, padded to ITS longest
    target."""
    hp = _hp()
    root = _corpus(tmp_path, 40)
    per_rank = []
    for rank in range(2):
        loader, features, _ = _stubs(hp)
        f = DataFeeder(hp, ljspeech=root, seed=9, rank=rank, world=2, prefetch=False, features=features, loader=loader)
        per_rank.append([f.next_batch() for _ in range(6)])           # two groups
    for (_, _, m0, l0), (_, _, m1, l1) in zip(*per_rank):
        assert m0.shape[1] == m1.shape[1] == l0.shape[1] == l1.shape[1]
        longest = max(int(round(m[:, :, 0].sum(axis=1).max() / 0.25)) for m in (m0, m1))     # the stub's mel is 0.25
        r = hp.outputs_per_step
        assert m0.shape[1] == (longest + 1 + r - 1) // r * r


def test_prefetch_thread_binds_its_device(tmp_path, monkeypatch):
    """torch's current device is per host thread: the worker must select the rank's GPU itself before the feature
    kernels run (train.py calls set_device only in the main thread)."""
    import threading

    import torch
    seen = []
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: seen.append((threading.current_thread().name, d)))
    hp = _hp()
    root = _corpus(tmp_path, 12)
    loader, features, _ = _stubs(hp)
    f = DataFeeder(hp, ljspeech=root, seed=1, prefetch=True, features=features, loader=loader, device=1)
    f.next_batch()
    assert ("datafeeder", 1) in seen


def test_cmudict_substitution(tmp_path):
    hp = _hp()
    root = _corpus(tmp_path, 12)
    d = cmudict.CMUDict(__import__("io").StringIO("HELLO  HH AH0 L OW1\nWORLD  W ER1 L D\nSTREET  S T R IY1 T\n"))
    loader, features, _ = _stubs(hp)
    f = DataFeeder(hp, ljspeech=root, seed=1, prefetch=False, cmudict=d, features=features, loader=loader)
    texts = []
    for _ in range(12):
        inputs, lengths, _, _ = f.next_batch()
        texts += [sequence_to_text(inputs[i, :lengths[i]]) for i in range(len(lengths))]
    assert any("{HH AH0 L OW1}" in t for t in texts) and any("hello" in t for t in texts)
    assert all(t.endswith("~") for t in texts)


def test_multi_corpus_speaker_numbering(tmp_path):
    """datafeeder.py:44-61,166-167: items of every corpus, (corpus, speaker) pairs numbered once, ids travel with
    their utterances through sorting, dealing and the in-batch shuffle.  corpus/vctk.py:11-20 and
    corpus/ljspeech.py:14-27 give the directory conventions."""
    lj = tmp_path / "lj"
    os.makedirs(lj)
    _corpus(lj, 6)
    vctk = tmp_path / "vctk"
    for spk, n in (("225", 3), ("301", 2)):
        os.makedirs(vctk / "wav48" / ("p" + spk))
        os.makedirs(vctk / "txt" / ("p" + spk))
        for i in range(n):
            open(vctk / "wav48" / ("p" + spk) / ("p%s_%03d.wav" % (spk, i)), "wb").close()
            with open(vctk / "txt" / ("p" + spk) / ("p%s_%03d.txt" % (spk, i)), "w") as f:
                f.write("speaker %s says hello\n" % ("a" if spk == "225" else "b"))
    open(vctk / "wav48" / "p225" / "p225_099.wav", "wb").close()       # no transcript: skipped
    libre = tmp_path / "libre"
    os.makedirs(libre)
    with open(libre / "corpus.csv", "w") as f:
        f.write("1272-128104-0012,dev-clean/1272/128104/1272-128104-0012.wav,only his own work,training\n")
        f.write("84-121123-0001,dev-clean/84/121123/84-121123-0001.wav,go on,training\n")
    hp = _hp(batch_size=2, batch_group_size=2)
    seen = {}

    def loader(path):
        return path

    def features(path):
        T = 20 + len(os.path.basename(path)) % 7
        seen[path] = T
        return np.full((hp.num_freq, T), 0.5, np.float32), np.full((hp.num_mels, T), 0.25, np.float32)
    fd = DataFeeder(hp, ljspeech=str(lj), vctk=str(vctk), librispeech=str(libre), seed=2, prefetch=False,
                    features=features, loader=loader, trim=False)
    assert len(fd.items) == 6 + 5 + 2
    assert fd.id2speaker == {0: ("libre", "1272"), 1: ("libre", "84"), 2: ("ljspeech", "0"), 3: ("vctk", "225"),
                             4: ("vctk", "301")}
    # the same numbering on another rank (sorted pairs, no dependence on set order)
    fd1 = DataFeeder(hp, ljspeech=str(lj), vctk=str(vctk), librispeech=str(libre), seed=2, rank=1, world=2,
                     prefetch=False, features=features, loader=loader, trim=False)
    assert fd1.speaker2id == fd.speaker2id
    # ids follow their utterances: VCTK texts name their speaker, LJSpeech rows are speaker 2
    count = {}
    for _ in range(8):
        inputs, lengths, mel, lin = fd.next_batch()
        assert fd.speaker_ids.shape == (2,) and fd.speaker_ids.dtype == np.int32
        for i in range(2):
            text = sequence_to_text(inputs[i, :lengths[i]])
            sid = int(fd.speaker_ids[i])
            count[sid] = count.get(sid, 0) + 1
            if text.startswith("speaker a"):
                assert sid == 3
            elif text.startswith("speaker b"):
                assert sid == 4
            elif text.startswith("only his"):
                assert sid == 0
            elif text.startswith("go on"):
                assert sid == 1
            else:
                assert sid == 2
    assert set(count) == {0, 1, 2, 3, 4}


# ---------------------------------------------------------------------------------------------------------------------
# silence trimming in front of the features (process.py:27,39-42,56-68)

def _utterance(lead, body, tail, seed, noise=1e-4, amp=0.5):
    rng = np.random.default_rng(seed)
    w = rng.normal(0, noise, lead + body + tail)
    t = np.arange(body) / 20000.0
    w[lead:lead + body] += amp * np.sin(2 * np.pi * 180 * t) * (1 + 0.3 * np.sin(2 * np.pi * 3 * t))
    return w.astype(np.float32)


def test_trim_wav_matches_the_oracle_frame_loops():
    """The product's running-sum frame energies against the oracle's per-frame loops (librosa 0.6.0 effects.split restated,
    [3P]): both margins, no silence at all, all silence, a burst too short to count, two bursts with a pause, a quiet
    stretch just above / below the 25 dB line, and lengths around the frame grid."""
    from oracle import audio_oracle as AO
    from nspeech_amd.datasets import process as P
    cases = [_utterance(6000, 40000, 9000, 0), _utterance(0, 30000, 0, 1), _utterance(500, 30000, 700, 2),
             _utterance(3000, 1500, 3000, 3), _utterance(8000, 300, 8000, 4), np.zeros(5000, np.float32),
             _utterance(1024, 5000, 511, 5), _utterance(1023, 5001, 513, 6), _utterance(2500, 2049, 2500, 7)]
    two = _utterance(4000, 12000, 3000, 8)
    cases.append(np.concatenate([two, _utterance(5000, 9000, 6000, 9)]))
    for db in (-24.0, -26.0):          # a tail 24 / 26 dB below the loudest frame: kept / cut
        w = _utterance(3000, 20000, 0, 10)
        tail = _utterance(0, 6000, 4000, 11, amp=0.5 * 10 ** (db / 20))
        cases.append(np.concatenate([w, tail]))
    lens = []
    for w in cases:
        got_s = P.split(w, 25, frame_length=1024, hop_length=512)
        ref_s = AO.effects_split(w, 25, 1024, 512)
        assert [tuple(int(v) for v in r) for r in got_s] == ref_s
        got, ref = P.trim_wav(w), AO.trim_wav(w)
        assert got.dtype == w.dtype and np.array_equal(got, ref)
        lens.append((len(w), len(got)))
    assert lens[0] == (55000, 44960)                       # [5632, 46592) widened by 2000 on both sides
    assert lens[1][0] == lens[1][1] and lens[2][0] == lens[2][1]
    assert lens[3] == (7500, 6560)                         # a 1500-sample burst covers 5 frames = 2560 > 2000 samples
    assert lens[4][0] == lens[4][1]                        # a 300-sample burst covers <= 2000 samples: nothing is cut
    assert lens[5] == (5000, 5000)                         # all silence: every frame sits at the 1e-10 floor = "loud"
    assert lens[-2][1] > lens[-1][1]                       # the -24 dB tail stays, the -26 dB tail goes


def test_find_start_and_end_follow_the_reference_rules():
    from nspeech_amd.datasets.process import _find_end, _find_start
    splits = np.array([[100, 900], [5000, 9000], [12000, 12500], [20000, 30000]])
    assert _find_start(splits) == 3000 and _find_end(splits, 31000) == 31000 and _find_end(splits, 40000) == 32000
    assert _find_start(np.array([[500, 4000]])) == 0                       # clipped at the first sample
    assert _find_start(np.zeros((0, 2), np.int64)) == 0 and _find_end(np.zeros((0, 2), np.int64), 77) == 77
    assert _find_start(np.array([[0, 2000]])) == 0 and _find_end(np.array([[0, 2000]]), 5000) == 5000   # needs > 2000


def test_feeder_takes_its_features_from_the_trimmed_wav(tmp_path):
    """process.py:27: the feeder's features are those of trim_wav(load_wav(path)), not of the file."""
    from oracle import audio_oracle as AO
    hp = _hp(batch_size=2, batch_group_size=1)
    root = _corpus(tmp_path, 2)
    wavs = {0: _utterance(7000, 30000, 6000, 20), 1: _utterance(0, 26000, 9000, 21)}
    seen = []

    def loader(path):
        return wavs[int(os.path.basename(path)[3:6])]

    def features(wav):
        seen.append(np.array(wav))
        T = 1 + len(wav) // 250
        return np.full((hp.num_freq, T), 0.5, np.float32), np.full((hp.num_mels, T), 0.25, np.float32)
    f = DataFeeder(hp, ljspeech=root, seed=1, prefetch=False, features=features, loader=loader)
    _, _, mel, _ = f.next_batch()
    want = [AO.trim_wav(wavs[i]) for i in (0, 1)]
    assert sorted(len(s) for s in seen) == sorted(len(w) for w in want)
    for s in seen:
        assert any(np.array_equal(s, w) for w in want)
    assert all(len(w) < len(wavs[i]) for i, w in enumerate(want))
    assert sorted(int(v) for v in mel[:, :, 0].sum(axis=1) * 4) == sorted(1 + len(w) // 250 for w in want)


def test_stop_ends_the_background_thread_and_start_resumes(tmp_path):
    """DataFeeder.stop(): the thread ends, nothing stays queued; next_batch() afterwards starts a new thread."""
    import threading
    hp = _hp()
    root = _corpus(tmp_path, 24)
    loader, features, _ = _stubs(hp)
    f = DataFeeder(hp, ljspeech=root, seed=3, features=features, loader=loader).start()
    a = f.next_batch()
    before = threading.active_count()
    f.stop()
    assert f._thread is None and f._queue.empty() and threading.active_count() == before - 1
    b = f.next_batch()
    assert b[0].shape[0] == a[0].shape[0]
    f.stop()
