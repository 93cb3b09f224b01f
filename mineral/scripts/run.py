"""`python -m mineral.scripts.run ...` -> the MI355X launcher."""

from mineral_amd.scripts.run import cli, main, make_datasets, make_envs  # noqa: F401

if __name__ == '__main__':
    cli()
