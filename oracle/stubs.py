"""TEST INFRASTRUCTURE (oracle).  Import shims that let the UNMODIFIED reference trainers under
/root/reference be imported in a container that lacks pytorch_lightning / torchrec / tensorboard.

Only oracle/reference_harness.py, oracle/make_golden.py and oracle/build_ref.py use this where /root/reference exists
(the build container), and bench.py's cpu_baseline leg where only the byte-compiled oracle/_ref exists (the GPU box).
Nothing under reagent_amd/ imports it.
The recipe is the one recorded in SURVEY.md §8(c).
"""
import enum
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("REAGENT_REFERENCE_ROOT", "/root/reference")
# the same reference byte-compiled by oracle/build_ref.py (git-ignored; travels to the GPU box with the snapshot)
BUILT_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def reference_available() -> bool:
    """the reference SOURCE tree is here (build container): what the golden-fixture tests and generators need"""
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "reagent"))


def runtime_root():
    """where `install()` imports `reagent` from: the source tree when present, else the byte-compiled `oracle/_ref`
    (GPU box: bench.py's cpu_baseline leg), else None"""
    if reference_available():
        return REFERENCE_ROOT
    if os.path.isfile(os.path.join(BUILT_ROOT, "reagent", "__init__.pyc")):
        return BUILT_ROOT
    return None


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Anything:
    """Class whose every attribute / call is a harmless no-op."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything()


def install():
    """Register the fake third-party modules and a bypass package for reagent.training."""
    if "reagent.training" in sys.modules and getattr(sys.modules["reagent.training"], "_oracle_stub", False):
        return
    import torch
    import torch.nn as nn

    # ---- torchrec (import-time only: reagent/core/types.py:22-23 etc.) ----
    class PoolingType(enum.Enum):
        SUM = "SUM"
        MEAN = "MEAN"
        NONE = "NONE"

    cls = lambda n: type(n, (), {"__init__": lambda self, *a, **k: None})  # noqa: E731
    _mod("torchrec", PoolingType=PoolingType, EmbeddingBagCollection=cls("EmbeddingBagCollection"),
         EmbeddingBagConfig=cls("EmbeddingBagConfig"))
    _mod("torchrec.sparse")
    _mod("torchrec.sparse.jagged_tensor", KeyedJaggedTensor=cls("KeyedJaggedTensor"),
         JaggedTensor=cls("JaggedTensor"))
    _mod("torchrec.models")
    _mod("torchrec.models.dlrm", SparseArch=cls("SparseArch"), InteractionArch=cls("InteractionArch"))
    _mod("torchrec.modules")
    _mod("torchrec.modules.embedding_modules", EmbeddingBagCollection=cls("EmbeddingBagCollection"))
    _mod("torchrec.modules.embedding_configs", EmbeddingBagConfig=cls("EmbeddingBagConfig"),
         PoolingType=PoolingType)
    _mod("torchrec.metrics")
    _mod("torchrec.metrics.metric_module")

    # ---- pytorch_lightning 1.6 surface used by reagent_lightning_module.py ----
    class LightningModule(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self.logger = None
            self.trainer = None
            self.current_epoch = 0
            self.on_gpu = False

        def log(self, *a, **k):
            pass

        def optimizers(self, use_pl_optimizer=True):
            return []

    class DummyExperiment:
        def nop(self, *a, **k):
            pass

        def __getattr__(self, _):
            return self.nop

        def __getitem__(self, idx):
            return self

    pl = _mod("pytorch_lightning", LightningModule=LightningModule, Callback=cls("Callback"),
              LightningDataModule=cls("LightningDataModule"), Trainer=cls("Trainer"),
              seed_everything=lambda *a, **k: None)
    loggers = _mod("pytorch_lightning.loggers")
    base = _mod("pytorch_lightning.loggers.base", DummyExperiment=DummyExperiment,
                LoggerCollection=cls("LoggerCollection"), LightningLoggerBase=cls("LightningLoggerBase"),
                rank_zero_experiment=lambda f: f)
    tb = _mod("pytorch_lightning.loggers.tensorboard", TensorBoardLogger=cls("TensorBoardLogger"))
    util = _mod("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
    pl.loggers, loggers.base, loggers.tensorboard, pl.utilities = loggers, base, tb, util
    loggers.TensorBoardLogger = tb.TensorBoardLogger

    # ---- tensorboard (torch.utils.tensorboard -> reagent/core/tensorboardX.py:26) ----
    if "torch.utils.tensorboard" not in sys.modules:
        try:
            importlib.import_module("torch.utils.tensorboard")
        except Exception:
            tbm = _mod("torch.utils.tensorboard", SummaryWriter=_Anything)
            torch.utils.tensorboard = tbm

    # ---- reference on sys.path; bypass reagent/training/__init__.py (imports ~17 trainers) ----
    root = runtime_root()
    if root is None:
        raise RuntimeError("oracle.stubs: neither the reference tree nor oracle/_ref (python -m oracle.build_ref) is here")
    if root not in sys.path:
        sys.path.insert(0, root)
    importlib.import_module("reagent")
    tr = types.ModuleType("reagent.training")
    tr.__path__ = [os.path.join(root, "reagent", "training")]
    tr._oracle_stub = True
    sys.modules["reagent.training"] = tr


def install_gym():
    """A bare `gym` module plus bypass packages for reagent.gym / reagent.gym.preprocessors, so that the
    unmodified input makers (reagent/gym/preprocessors/trainer_preprocessor.py:100-227) import without the
    gym package (absent here) and without reagent/gym/__init__.py (which imports every environment)."""
    install()
    if getattr(sys.modules.get("reagent.gym"), "_oracle_stub", False):
        return
    spaces = _mod("gym.spaces", Discrete=type("Discrete", (), {}), Box=type("Box", (), {}), Dict=type("Dict", (), {}))
    _mod("gym", Env=type("Env", (), {}), spaces=spaces)
    for name, sub in (("reagent.gym", "gym"), ("reagent.gym.preprocessors", os.path.join("gym", "preprocessors")),
                      ("reagent.gym.datasets", os.path.join("gym", "datasets"))):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(runtime_root(), "reagent", sub)]
        m._oracle_stub = True
        sys.modules[name] = m
    # reagent/gym/datasets/replay_buffer_dataset.py:10-15 imports Agent / EnvWrapper (annotations only: their own modules
    # import every environment and policy) and the two factories from the preprocessors PACKAGE (whose __init__ is bypassed)
    _mod("reagent.gym.agents")
    _mod("reagent.gym.agents.agent", Agent=type("Agent", (), {}))
    _mod("reagent.gym.envs", EnvWrapper=type("EnvWrapper", (), {}))
    pre = sys.modules["reagent.gym.preprocessors"]
    ins = importlib.import_module("reagent.gym.preprocessors.replay_buffer_inserters")
    tp = importlib.import_module("reagent.gym.preprocessors.trainer_preprocessor")
    pre.make_replay_buffer_inserter = ins.make_replay_buffer_inserter
    pre.make_replay_buffer_trainer_preprocessor = tp.make_replay_buffer_trainer_preprocessor
