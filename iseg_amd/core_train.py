"""core_train.py of the reference (:22-204): CoreTrain -- strategy scope -> compiled model -> shuffle/repeat/batch/prefetch ->
fit with checkpoint / model / time callbacks."""
import math

from . import dist
from . import nn
from .callbacks import CheckpointSaver, ModelCallback, TimeCallback
from .utils.model_utils import create_compiled_model


class CoreTrain(object):
    def __init__(self, model_helper, train_dataset, val_dataset=None, val_image_count=0, use_tpu=False, use_tpu_pod=False,
                 use_data_shared_policy_for_train=True, use_data_shared_policy_for_val=True):
        self.model_helper = model_helper
        self.training_dataset = train_dataset
        self.val_dataset = val_dataset
        self.val_image_count = val_image_count
        self.use_tpu = use_tpu
        self.use_tpu_pod = use_tpu_pod
        self.use_data_shared_policy_for_train = use_data_shared_policy_for_train
        self.use_data_shared_policy_for_val = use_data_shared_policy_for_val

    def create_trainable_model(self, num_class, ignore_label=255, class_weights=None, batch_size=1, epoch_steps=1000, initial_epoch=0,
                               jit_compile=None):
        return create_compiled_model(model=self.model_helper.model, num_class=num_class, ignore_label=ignore_label,
                                     class_weights=class_weights, batch_size=batch_size, epoch_steps=epoch_steps,
                                     initial_epoch=initial_epoch, jit_compile=jit_compile, optimizer=self.model_helper.optimizer)

    def train(self, distribute_strategy, num_class=21, ignore_label=255, class_weights=None, batch_size=1, eval_batch_size=None,
              shuffle_rate=100, epoch_steps=1000, initial_epoch=0, train_epoches=30, tensorboard_dir="tensorboard", use_profiler=False,
              verbose=1, validation_freq=1, jit_compile=None):
        if eval_batch_size is None:
            eval_batch_size = batch_size
        with distribute_strategy.scope():
            model = self.create_trainable_model(num_class, ignore_label=ignore_label, class_weights=class_weights,
                                                batch_size=batch_size, epoch_steps=epoch_steps, initial_epoch=initial_epoch,
                                                jit_compile=jit_compile)
        if initial_epoch == -1:
            initial_epoch = model.optimizer.iterations // epoch_steps
        # `batch_size` is the GLOBAL batch (MirroredStrategy splits it evenly over the replicas)
        replicas = distribute_strategy.num_replicas_in_sync
        per_replica = max(batch_size // replicas, 1)
        train_ds = self.prepare_train_dataset(model, per_replica, shuffle_rate)
        eval_ds = self.prepare_val_dataset(model, max(eval_batch_size // replicas, 1))
        val_steps = None if eval_ds is None else int(math.ceil(self.val_image_count / eval_batch_size))
        callbacks = [CheckpointSaver(self.model_helper), ModelCallback(self.model_helper.model), TimeCallback()]
        return model.fit(train_ds, epochs=train_epoches, validation_data=eval_ds, callbacks=callbacks, initial_epoch=initial_epoch,
                         steps_per_epoch=epoch_steps, validation_steps=val_steps, verbose=verbose, validation_freq=validation_freq)

    def prepare_train_dataset(self, model, batch_size=1, shuffle_rate=100):
        ds = self.handle_custom_dataprocess(self.training_dataset, model)
        if dist.world_size() > 1:
            ds = ds.shard(dist.world_size(), dist.rank())
        ds = ds.shuffle(shuffle_rate)
        ds = ds.repeat()
        ds = ds.batch(batch_size, drop_remainder=self.use_tpu)
        ds = ds.prefetch(buffer_size=2, device=nn.device())
        return ds

    def prepare_val_dataset(self, model, batch_size=1):
        if self.val_dataset is None:
            return None
        ds = self.handle_custom_dataprocess(self.val_dataset, model)
        if dist.world_size() > 1:
            ds = ds.shard(dist.world_size(), dist.rank())
        ds = ds.repeat()
        ds = ds.batch(batch_size, drop_remainder=self.use_tpu)
        ds = ds.prefetch(buffer_size=2, device=nn.device())
        return ds

    def data_based_shard_policy(self, ds, use_data_shared_policy=True):
        return ds

    def handle_custom_dataprocess(self, ds, model):
        fn = getattr(model.model if hasattr(model, "model") else model, "inputs_process", None)
        if fn is not None and callable(fn):
            ds = ds.map(fn)
        return ds
