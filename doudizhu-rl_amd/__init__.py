"""doudizhu-rl_amd: MI355X-native batched Doudizhu environment (hot path of
charleschen003/doudizhu-rl: deal, legal-move enumeration, action application,
terminal/reward, state encoding).  HIP kernels + C ABI under csrc/, host mirror of the
reference's envi.py here.  Import never touches the GPU; using the engine without the
built library or without an MI355X raises (there is no CPU fallback)."""
from ._lib import DdzError  # noqa: F401
from .engine import (BatchedEnv, FACE_PLANES, NUM_ACTIONS, STEP_CHOICE, STEP_IDS, STEP_RANDOM,  # noqa: F401
                     STEP_ROWS, TRAJ_BYTES, TRAJ_PACKED_BYTES, action_table, auto_choose, cards_value, get_moves, get_moves_slab,
                     pack_trajectory, q_features, q_features_packed, q_features_needed, q_fc1_dense, q_fc1_rows, q_fc1_rows_k,
                     q_features_rows, q_features_drows, q_gather_h0, q_shared_ws_bytes, q_shared_need_ws_bytes,
                     rows_to_onehot, state_prob)
from .envi import Env, EnvComplicated, EnvCooperation, EnvCooperationSimplify  # noqa: F401
