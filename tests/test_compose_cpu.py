"""Host logic of the launcher: the hydra-subset config composer and the `mineral` drop-in namespace."""
import os
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mineral_amd.scripts import compose as C  # noqa: E402
from mineral_amd.scripts.utils import launcher_seeds  # noqa: E402

CFGS = os.path.join(ROOT, 'mineral_amd', 'cfgs')


def write(base, rel, text):
    path = os.path.join(base, rel)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w') as f:
        f.write(textwrap.dedent(text))


@pytest.fixture
def tree(tmp_path):
    base = str(tmp_path / 'cfgs')
    write(base, 'config.yaml', '''
        defaults:
          - task: A
          - agent: ${task}PPO
          - _self_
        hydra:
          searchpath: []
        task_name: ${task.name}
        num_envs: ''
        logdir: 'workdir/${task.name}'
        multi_gpu: False
        rl_device: ???
        seed: 42
    ''')
    write(base, 'task/A.yaml', '''
        suite: synthetic
        name: A
        env:
          numEnvs: ${resolve_default:64,${...num_envs}}
    ''')
    write(base, 'agent/Base.yaml', '''
        seed: ${..seed}
        algo: PPO
        network:
          units: [512, 256]
          keys: x
        ppo:
          multi_gpu: ${...multi_gpu}
          num_actors: ${...task.env.numEnvs}
          max_agent_steps: 4.1e6
          optim_kwargs: {lr: 5e-4, eps: 1e-5}
    ''')
    write(base, 'agent/Mixin.yaml', '''
        network:
          encoder: Enc
          encoder_kwargs:
            pcd_keys: '^q$'
            inputs: {pos_keys: '${..pcd_keys}'}
    ''')
    write(base, 'agent/APPO.yaml', '''
        defaults:
          - Base
          - Mixin
          - _self_
        network:
          keys: y
        ppo:
          extra: ${if:${eq:${...task.name},a},yes,no}
    ''')
    return base


def test_defaults_inheritance_and_relative_interpolation(tree):
    t = C.compose(tree, [])
    t['rl_device'] = 'cpu'
    c = C.resolve(t)
    assert c.agent.algo == 'PPO' and c.agent.seed == 42
    assert c.agent.network.units == [512, 256] and c.agent.network['keys'] == 'y'      # _self_ after the parents
    assert c.agent.network.encoder_kwargs.inputs.pos_keys == '^q$'                  # ${..pcd_keys}
    assert c.agent.ppo.num_actors == 64 and c.task.env.numEnvs == 64                # resolve_default with ''
    assert c.agent.ppo.multi_gpu is False
    assert c.agent.ppo.max_agent_steps == 4.1e6 and isinstance(c.agent.ppo.max_agent_steps, float)
    assert c.agent.ppo.optim_kwargs == {'lr': 5e-4, 'eps': 1e-5}                    # exponent floats without a dot
    assert c.logdir == 'workdir/A' and c.task_name == 'A'
    assert c.agent.ppo.extra == 'yes'                                               # nested resolvers


def test_overrides(tree):
    t = C.compose(tree, ['num_envs=4096', 'agent.ppo.max_agent_steps=1e5', '+agent.ppo.new_key=[1,2]', '~agent.network.encoder',
                         'seed=7', 'logdir=/tmp/x', 'multi_gpu=True'])
    t['rl_device'] = 'cuda:0'
    c = C.resolve(t)
    assert c.task.env.numEnvs == 4096 and c.agent.ppo.num_actors == 4096
    assert c.agent.ppo.max_agent_steps == 1e5 and c.agent.ppo.new_key == [1, 2]
    assert 'encoder' not in c.agent.network and c.agent.seed == 7 and c.logdir == '/tmp/x'
    assert c.agent.ppo.multi_gpu is True
    with pytest.raises(C.ComposeError):
        C.compose(tree, ['agent.ppo.typo=1'])          # overriding a key that does not exist needs '+'
    with pytest.raises(C.ComposeError):
        C.compose(tree, ['agent=Nope'])


def test_missing_value_and_searchpath(tree, tmp_path):
    with pytest.raises(C.ComposeError, match='rl_device'):
        C.resolve(C.compose(tree, []))                   # '???' left unset
    ext = str(tmp_path / 'ext')
    write(ext, 'task/B.yaml', 'suite: myenv\nname: B\nenv: {numEnvs: 3}\n')
    write(ext, 'agent/BPPO.yaml', 'algo: MyAgent\n')
    t = C.compose(tree, [f'hydra.searchpath=[file://{ext}]', 'task=B'])
    t['rl_device'] = 'cpu'
    c = C.resolve(t)
    assert c.task.suite == 'myenv' and c.agent.algo == 'MyAgent' and c.task_name == 'B'


def test_shipped_configs_compose():
    t = C.compose(CFGS, ['num_envs=128'])
    t['sim_device'] = t['rl_device'] = 'cuda:0'
    c = C.resolve(t)
    assert c.task.suite == 'synthetic' and c.agent.algo == 'PPO'
    p = c.agent.ppo
    assert (p.horizon_len, p.minibatch_size, p.mini_epochs, p.optim_type, p.kl_threshold, p.max_grad_norm) == (32, 32768, 5, 'AdamW', 0.008, 0.5)
    assert p.num_actors == 128 and c.agent.network.actor_critic_kwargs.mlp_kwargs.units == [512, 256, 128]
    t = C.compose(CFGS, ['task=DFlex', 'agent=AllegroHandPPO'])
    t['sim_device'] = t['rl_device'] = 'cpu'
    c = C.resolve(t)
    assert (c.agent.ppo.optim_type, c.agent.ppo.horizon_len, c.agent.ppo.reward_shaper.scale, c.agent.ppo.kl_threshold) == ('Adam', 8, 0.01, 0.016)
    assert c.task.env.numEnvs == 64 and c.agent.ppo.minibatch_size == 32768


@pytest.mark.skipif(not os.path.isdir('/root/reference/mineral/cfgs'), reason='reference config tree not present')
def test_reference_config_tree_composes():
    """Every agent file of the reference composes and resolves (its yaml files load unchanged)."""
    base = '/root/reference/mineral/cfgs'
    for name in sorted(os.listdir(os.path.join(base, 'agent'))):
        agent = name[:-5]
        if agent == 'PointNet':
            continue                                     # a mixin, not an agent
        task = 'Rewarped' if agent.startswith('Rewarped') else 'DFlex'
        t = C.compose(base, [f'task={task}', f'agent={agent}', 'num_envs=16'])
        t['sim_device'] = t['rl_device'] = 'cpu'
        c = C.resolve(t)
        assert c.agent.algo in ('PPO', 'SAC', 'DDPG', 'SHAC', 'BPTT') and c.task.env.numEnvs == 16


def test_launcher_seeds_and_namespace():
    assert launcher_seeds(42) == {'setup': 121958, 'train': 671155, 'eval': 131932, 'env_train': 365838, 'env_eval': 259178}
    assert launcher_seeds(100)['setup'] == 824840
    import mineral
    import mineral.agents
    import mineral.envs
    import mineral_amd.agents
    assert mineral.agents is mineral_amd.agents and mineral.agents.PPO is mineral_amd.agents.PPO
    assert hasattr(mineral.envs, 'synthetic') and hasattr(mineral.envs, 'dflex')
    from mineral.scripts import run
    assert callable(run.cli) and callable(run.make_envs)


def test_prepare_picks_cpu_without_a_gpu(monkeypatch):
    import torch
    from mineral_amd.scripts import run
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: False)
    t = C.compose(CFGS, ['num_envs=8'])
    assert run.prepare(t) == 0
    c = C.resolve(t)
    assert c.rl_device == 'cpu' and c.sim_device == 'cpu' and c.device_id == -1
