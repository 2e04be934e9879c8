"""Host logic of train.py that needs no GPU: the checkpoint retention rule of the reference's
tf.train.Saver(max_to_keep=5, keep_checkpoint_every_n_hours=2) (train.py:60)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Model(object):
    def state_dict(self):
        return {"global_step": 0}


def test_checkpoint_retention_keeps_five_and_one_every_two_hours(tmp_path):
    import train
    now = [1000.0]
    saver = train.CheckpointSaver(str(tmp_path), clock=lambda: now[0])
    # a checkpoint every 30 minutes for 8 hours
    for i in range(1, 17):
        now[0] = 1000.0 + i * 1800.0
        saver.save(_Model(), i * 1000)
    have = sorted(int(n.split("-")[-1]) for n in os.listdir(tmp_path) if n.startswith("model.ckpt-"))
    # the newest five, plus the first file written later than each 2-hour mark (marks at 2 h, 4 h, ... after the saver
    # was created; a file is only judged when it leaves the window of five)
    assert have[-5:] == [12000, 13000, 14000, 15000, 16000]
    assert have[:-5] == [5000, 9000], have        # written at 2.5 h (> 2 h mark) and 4.5 h (> 4 h mark)
    with open(os.path.join(tmp_path, "checkpoint")) as f:
        assert f.read().strip().splitlines()[-1] == 'model_checkpoint_path: "model.ckpt-16000"'


def test_checkpoint_retention_plain_window(tmp_path):
    import train
    now = [0.0]
    saver = train.CheckpointSaver(str(tmp_path), clock=lambda: now[0])
    for i in range(1, 9):
        now[0] += 60.0                                   # eight checkpoints within minutes: nothing is kept for good
        saver.save(_Model(), i)
    have = sorted(int(n.split("-")[-1]) for n in os.listdir(tmp_path) if n.startswith("model.ckpt-"))
    assert have == [4, 5, 6, 7, 8]
