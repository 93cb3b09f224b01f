"""Launcher: `python -m mineral.scripts.run task=... agent=... key=value` (mineral/scripts/run.py)."""
