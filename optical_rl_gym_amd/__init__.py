"""optical_rl_gym_amd: batched RWA / RMSA / DeepRMSA / RMCSA optical-network environments whose
reset()/step()/heuristic hot path runs as hand-written HIP kernels on AMD MI355X (gfx950)."""
from .envs import (BatchedDeepRMSAEnv, BatchedOpticalEnv, BatchedQoSConstrainedRA, BatchedRMCSAEnv,  # noqa: F401
                   BatchedRMSAEnv, BatchedRWAEnv, make)
from .topology import Modulation, Path, Topology, get_best_modulation_format  # noqa: F401

__version__ = "0.1.0"
from .gym_api import (DeepRMSAEnv, PathOnlyFirstFitAction, RMCSAEnv, RMSAEnv, RWAEnv, Service,  # noqa: F401,E402
                      SimpleMatrixObservation, evaluate_heuristic, least_loaded_path_first_fit, random_policy,
                      start_environment,
                      shortest_available_path_best_modulation_first_core_first_fit,
                      shortest_available_path_first_fit, shortest_available_path_last_fit, shortest_path_first_fit)
from .sharding import MultiDeviceBatch, shard_range, shard_seeds  # noqa: F401,E402
from .vec_env import OpticalVecEnv  # noqa: F401,E402
from .qos import QoSConstrainedRA  # noqa: F401,E402
