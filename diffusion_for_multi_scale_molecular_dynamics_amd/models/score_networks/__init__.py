from .score_network import ScoreNetwork, ScoreNetworkParameters  # noqa: F401
