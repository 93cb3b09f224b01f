"""`mineral` — drop-in name for the MI355X build.

Putting this repository on PYTHONPATH makes `python -m mineral.scripts.run task=... agent=...` and
`import mineral.agents / mineral.envs / mineral.common / mineral.scripts` resolve to `mineral_amd`, so launch lines,
plugin registrations (`mineral.agents.MyAgent = MyAgent`, `mineral.envs.myenv = myenv`) and imports written for the
reference keep working for the parts this build covers.  Anything it does not cover raises ImportError / AttributeError
on use.
"""

import importlib
import sys

import mineral_amd

__version__ = mineral_amd.__version__

for _name in ('agents', 'agents.ppo', 'agents.sac', 'agents.diffrl', 'buffers', 'common', 'envs', 'scripts', 'config', 'parallel'):
    _mod = importlib.import_module(f'mineral_amd.{_name}')
    sys.modules[f'{__name__}.{_name}'] = _mod
    if '.' not in _name:
        globals()[_name] = _mod
del _name, _mod
