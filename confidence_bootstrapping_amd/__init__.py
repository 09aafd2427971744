from .hetero import HeteroData, Batch, DataLoader  # noqa: F401
