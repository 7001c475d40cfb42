from .temporal_graph import TemporalGraphAug  # noqa: F401
