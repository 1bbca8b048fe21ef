from .offline_table import OfflineTable  # noqa: F401
