"""Constants of the reference's config.py:8-18, copied by value (the reference module also
creates log directories and imports pytz at import time, which the hot path does not need)."""
import torch

GAMMA = 0.95
EPSILON_HIGH = 0.5
EPSILON_LOW = 0.01
REPLAY_SIZE = 20000
BATCH_SIZE = 256
DECAY = int((8000 * (2 / 3)) / 5)
UPDATE_TARGET_EVERY = 20

CARDS = range(3, 18)
STR = [str(i) for i in range(3, 11)] + ['J', 'Q', 'K', 'A', '2', '小', '大']
DICT = dict(zip(CARDS, STR))

DEVICE = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
