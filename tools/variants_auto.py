VARIANTS = {"r1": [], "r2": [], "r4": [], "r99": []}
FLAGS = {"r1": ["-DA2_SCAN_ROUNDS=1"], "r2": ["-DA2_SCAN_ROUNDS=2"], "r4": ["-DA2_SCAN_ROUNDS=4"], "r99": ["-DA2_SCAN_ROUNDS=99"]}
