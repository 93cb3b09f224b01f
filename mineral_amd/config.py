"""Attribute-style config tree with the access patterns the agents use on a hydra DictConfig:
`cfg.a.b`, `cfg['a']`, `cfg.get('a', default)` (mineral/agents/ppo/ppo.py:23-26,54-58,63-82)."""


class Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(d):
        if isinstance(d, Config):
            return d
        if isinstance(d, dict):
            return Config({k: Config.wrap(v) for k, v in d.items()})
        if isinstance(d, (list, tuple)):
            return type(d)(Config.wrap(v) for v in d)
        return d

    def to_container(self):
        def plain(x):
            if isinstance(x, dict):
                return {k: plain(v) for k, v in x.items()}
            if isinstance(x, (list, tuple)):
                return [plain(v) for v in x]
            return x
        return plain(self)
