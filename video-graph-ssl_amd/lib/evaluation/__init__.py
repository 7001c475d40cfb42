from .metric import AverageMeter, accuracy, accuracy_from_rank, rank_ge  # noqa: F401
