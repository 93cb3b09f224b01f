"""Device-resident replay buffers with the interfaces of mineral/buffers (ReplayBuffer, NStepReplay)."""

from .nstep_replay import NStepReplay
from .simple_replay import ReplayBuffer, create_buffer
