"""callbacks/ of the reference: CheckpointSaver (ckpt_saver.py:11-26), ModelCallback (model_callback.py:11), TimeCallback
(time_callback.py:10-25)."""
import time


class Callback:
    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass


class CheckpointSaver(Callback):
    def __init__(self, model_helper):
        self.model_helper = model_helper

    def on_epoch_end(self, epoch, logs=None):
        if self.model_helper.checkpoint_dir is not None:
            self.model_helper.save_checkpoint()


class ModelCallback(Callback):
    def __init__(self, model):
        self.model = model

    def on_epoch_end(self, epoch, logs=None):
        fn = getattr(self.model, "on_epoch_end", None)
        if callable(fn):
            fn(epoch, logs or {})


class TimeCallback(Callback):
    def on_epoch_begin(self, epoch, logs=None):
        self.t0 = time.time()
        print(f"Epoch {epoch} started at {time.strftime('%Y-%m-%d %H:%M:%S')}")

    def on_epoch_end(self, epoch, logs=None):
        print(f"Epoch {epoch} finished in {time.time() - self.t0:.1f} s")
