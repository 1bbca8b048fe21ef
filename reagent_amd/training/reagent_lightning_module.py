"""Stand-in for reagent/training/reagent_lightning_module.py:18-200.

The reference base class derives from pytorch_lightning.LightningModule; Lightning is a
control-plane dependency (loop, checkpoint, logging) outside the hot path and is not installed on
the MI355X image.  This class keeps the generator-per-batch protocol (`training_step` :108-133,
`soft_update_result` :76-81, `_num_optimizing_steps` :141-143), the reporter plumbing and the
registered buffers (so ``state_dict()`` keys match), on a plain ``nn.Module``.  When
pytorch_lightning is importable a maintainer can swap the base class.
"""
import inspect
import logging

import torch
import torch.nn as nn

logger = logging.getLogger(__name__)


class _NoOpReporter:
    """Every attribute is a callable doing nothing (pl.loggers.base.DummyExperiment equivalent)."""

    def nop(self, *args, **kw):
        pass

    def __getattr__(self, _):
        return self.nop


class ReAgentLightningModule(nn.Module):
    def __init__(self, automatic_optimization=True):
        super().__init__()
        self._automatic_optimization = automatic_optimization
        self._training_step_generator = None
        self._reporter = _NoOpReporter()
        self._verified_steps = False
        self.logger = None
        self.register_buffer("_next_stopping_epoch", None)
        self.register_buffer("_cleanly_stopped", None)
        self._next_stopping_epoch = torch.tensor([-1]).int()
        self._cleanly_stopped = torch.ones(1)
        self._setup_input_type()
        self.train_batches_processed_this_epoch = 0
        self.val_batches_processed_this_epoch = 0
        self.test_batches_processed_this_epoch = 0
        self.all_batches_processed = 0
        self._num_optimizing_steps_cache = None

    def _setup_input_type(self):
        self._training_batch_type = None
        sig = inspect.signature(self.train_step_gen)
        assert "training_batch" in sig.parameters
        annotation = sig.parameters["training_batch"].annotation
        if annotation == inspect.Parameter.empty:
            return
        if hasattr(annotation, "from_dict"):
            self._training_batch_type = annotation

    def set_reporter(self, reporter):
        if reporter is None:
            reporter = _NoOpReporter()
        self._reporter = reporter
        return self

    @property
    def reporter(self):
        return self._reporter

    def set_clean_stop(self, clean_stop: bool):
        self._cleanly_stopped[0] = int(clean_stop)

    def increase_next_stopping_epochs(self, num_epochs: int):
        self._next_stopping_epoch += num_epochs
        self.set_clean_stop(False)
        return self

    def train_step_gen(self, training_batch, batch_idx: int):
        raise NotImplementedError

    def soft_update_result(self) -> torch.Tensor:
        """A dummy loss to trigger soft-update (CPU, like the reference)."""
        one = torch.ones(1, requires_grad=True)
        return one + one

    def log(self, *args, **kwargs):  # LightningModule.log — metrics sink, no-op without Lightning
        pass

    def training_step(self, batch, batch_idx: int, optimizer_idx: int = 0):
        """The generator-per-batch protocol of reagent_lightning_module.py:108-130: call k of a batch returns the k-th loss
        `train_step_gen` yields; the call for the last optimizer closes the batch (and, once per module, checks that the
        generator has nothing left)."""
        n_opt = self._num_optimizing_steps
        assert optimizer_idx == 0 or n_opt > 1
        gen = self._training_step_generator
        if gen is None:  # first optimizer of a new batch
            if isinstance(batch, dict) and self._training_batch_type:
                batch = self._training_batch_type.from_dict(batch)
            gen = self._training_step_generator = self.train_step_gen(batch, batch_idx)
        loss = next(gen)
        if optimizer_idx != n_opt - 1:
            return loss
        if not self._verified_steps:
            self._require_exhausted(gen, n_opt)
        self._training_step_generator = None
        self.all_batches_processed += 1
        return loss

    def _require_exhausted(self, gen, n_opt: int):
        """one loss per optimizer, no more (checked on the first batch only, like the reference)"""
        leftover = object()
        if next(gen, leftover) is not leftover:
            raise RuntimeError("training_step_gen() yields too many times."
                               "The number of yields should match the number of optimizers,"
                               f" in this case {n_opt}")
        self._verified_steps = True

    @property
    def _num_optimizing_steps(self) -> int:
        if self._num_optimizing_steps_cache is None:
            self._num_optimizing_steps_cache = len(self.configure_optimizers())
        return self._num_optimizing_steps_cache

    def train(self, *args):
        if (len(args) == 0) or ((len(args) == 1) and (isinstance(args[0], bool))):
            return super().train(*args)
        raise NotImplementedError(
            "Method .train() is not used for ReAgent Lightning trainers. Please use .fit() method of the pl.Trainer instead"
        )
