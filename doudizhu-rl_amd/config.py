"""The two module-level names of the reference's config.py that the env adapter reads
(config.py:16-20): the rank -> label map used by cards2str, and the default torch device.
The DQN hyper-parameters, directories and logger of that module belong to the training
scripts, which are outside this package."""
import torch

# ranks 3..17 = 3..10, J, Q, K, A, 2, small joker, big joker (envi.py:122-124)
_FACES = {11: 'J', 12: 'Q', 13: 'K', 14: 'A', 15: '2', 16: '小', 17: '大'}
DICT = {rank: _FACES.get(rank, str(rank)) for rank in range(3, 18)}

DEVICE = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
