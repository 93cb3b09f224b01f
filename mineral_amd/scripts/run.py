"""`python -m mineral.scripts.run task=<T> agent=<A> key=value +key=value ...` on the MI355X build.

Same flow as mineral/scripts/run.py:44-204 — compose the config, pick devices, write run metadata, derive the
sub-seeds, build env / datasets / agent through the `envs` and `agents` namespaces, run train / eval / train_eval,
swallow exceptions after printing them (run.py:154-177) — with three differences: configs are composed by
`compose.py` (no hydra), multi-GPU is plain torch.distributed with one process per GPU (no accelerate wrapper), and
wandb is optional.
"""

import functools
import os
import pprint
import sys
import traceback

import yaml

from . import compose as C
from .utils import launcher_seeds, limit_threads, set_np_formatting, set_seed

CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'cfgs')


def make_envs(config):
    from .. import envs
    suite = config.task.get('suite', 'isaacgymenvs')
    if not hasattr(envs, suite):
        raise NotImplementedError(f"task.suite '{suite}' is not registered in {envs.__name__} "
                                  f"(register with `{envs.__name__}.{suite} = module_with_make_envs`)")
    return getattr(envs, suite).make_envs(config)


def make_datasets(config, env):
    if 'datasets' not in config.agent:
        return None
    from .. import envs
    return getattr(envs, config.task.get('suite', 'isaacgymenvs')).make_datasets(config, env)


def save_run_metadata(logdir, run_name, run_id, resolved_config):
    meta = {'logdir': logdir, 'run_name': run_name, 'run_id': run_id}
    with open(os.path.join(logdir, 'run_metadata.yaml'), 'w') as f:
        yaml.dump(meta, f, default_flow_style=False)
    with open(os.path.join(logdir, 'resolved_config.yaml'), 'w') as f:
        yaml.dump(resolved_config, f, default_flow_style=False)


def prepare(tree):
    """Device / rank fields that the launcher fills in before the config is resolved (run.py:59-97)."""
    import torch
    if not torch.cuda.is_available():
        tree['device_id'] = -1
    rank = 0
    if tree.get('multi_gpu'):
        import torch.distributed as dist
        rank = int(os.getenv('LOCAL_RANK', '0'))
        torch.cuda.set_device(rank)
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29500')
            dist.init_process_group('nccl', device_id=torch.device(f'cuda:{rank}'))
        dev = f'cuda:{rank}'
        tree['graphics_device_id'] = rank
    else:
        did = tree.get('device_id', 0)
        dev = f'cuda:{did}' if did >= 0 else 'cpu'
        if did >= 0:
            torch.cuda.set_device(did)   # the HIP library launches on the current device: it has to be the one that owns the tensors
        tree['graphics_device_id'] = did if did >= 0 else 0
    tree['sim_device'] = dev
    tree['rl_device'] = dev
    if tree.get('ckpt'):
        tree['ckpt'] = os.path.abspath(tree['ckpt'])
    return rank


def main(config, rank=0):
    set_np_formatting()
    limit_threads(1)
    assert config.seed >= 0
    seed_fn = functools.partial(set_seed, torch_deterministic=config.torch_deterministic)
    seed_fn(0)

    logdir = config.logdir
    os.makedirs(logdir, exist_ok=True)
    resolved = config.to_container()
    print(pprint.pformat(resolved, compact=True, indent=1), '\n')

    wandb_run = None
    if rank == 0:
        run_name, run_id = os.path.basename(os.path.normpath(logdir)), f'local-{os.getpid()}'
        if config.wandb.get('mode', 'disabled') != 'disabled':
            try:
                import wandb
                os.environ['WANDB_START_METHOD'] = 'thread'
                wcfg = dict(config.wandb.to_container())
                if wcfg.get('group') is None:
                    parts = os.path.normpath(logdir).split(os.sep)
                    wcfg['group'] = parts[-2] if len(parts) > 1 else parts[-1]
                wandb_run = wandb.init(**wcfg, dir=logdir, config=resolved)
                run_name, run_id = wandb_run.name, wandb_run.id
            except ImportError:
                print('wandb is not installed: logging to tb / metrics.jsonl only')
        save_run_metadata(logdir, run_name, run_id, resolved)

    seeds = launcher_seeds(config.seed)
    print(f'Base Seed: {config.seed}, Seeds: {seeds}')
    seed_fn(seeds['setup'] + rank)

    print('Making Envs')
    config.env_seed = seeds['env_train'] + rank
    env = make_envs(config)
    print('-' * 20)
    print(f'Env: {env}')
    datasets = make_datasets(config, env)
    print(f'Datasets: {datasets}')

    from .. import agents
    if not hasattr(agents, config.agent.algo):
        raise NotImplementedError(f"agent.algo '{config.agent.algo}' is not registered in {agents.__name__}")
    AgentCls = getattr(agents, config.agent.algo)
    print(f'AgentCls: {AgentCls}', '\n')
    agent = AgentCls(config, logdir=logdir, accelerator=None, datasets=datasets, env=env)
    if config.ckpt:
        print(f'Loading checkpoint: {config.ckpt}')
        agent.load(config.ckpt, ckpt_keys=config.ckpt_keys)

    print('-' * 20)
    print(f'Running: {config.run}')
    try:
        if config.run == 'train':
            seed_fn(seeds['train'] + rank)
            agent.train()
        elif config.run == 'eval':
            seed_fn(seeds['eval'] + rank)
            agent.eval()
        elif config.run == 'train_eval':
            seed_fn(seeds['train'] + rank)
            agent.train()
            agent.load(os.path.join(agent.ckpt_dir, 'final.pth'))
            seed_fn(seeds['eval'] + rank)
            agent.eval()
        else:
            raise NotImplementedError(config.run)
    except (KeyboardInterrupt, Exception) as e:   # the reference reports and carries on to the cleanup (run.py:171-177)
        traceback.print_exc()
        print(f'Exception: {e}')

    if hasattr(env, 'renderer') and config.task.env.get('render', False):
        env.renderer.save()
    if wandb_run is not None:
        import wandb
        wandb.finish()
    return agent


def cli(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    tree = C.compose(CONFIG_DIR, argv)
    rank = prepare(tree)
    config = C.resolve(tree)
    return main(config, rank)


if __name__ == '__main__':
    cli()
