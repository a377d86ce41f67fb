"""Signal-processing constants of the reference's model front end
(`/root/reference/constants.py:13-19`), imported as `c` like the reference does, plus the paths its
file-driven entry points read (`constants.py:3-10`: `evaluation.evaluate()` and
`model.create_speaker_models()` look under ROOT for `Models/model_14_percent_best_so_far.pt`,
`50_first_ids.txt`, `50_first_ids.npy`, `speaker_models/`, and under DATA_ORIGIN for the WAVs)."""
import os

ROOT = os.path.abspath(os.path.dirname(__file__))
DATA_TEMP = os.path.join(ROOT, 'data_temp/')
DATA_ORIGIN = os.path.join(ROOT, 'data_temp_small/')
NUM_FILES = 0                     # 0 = every listed file (load_data.py:42-45)
DERIVATIVE = False
NORMALIZE = False
SAMPLE_RATE = 16000
FRAME_LEN = 0.025
FRAME_STEP = 0.01
NUM_COEF = 40
NUM_FFT = 1024
BATCH_SIZE = 32
# FeatureCube((80, 40, 20)) in utils.py:20-21
CUBE_FRAMES = 80
CUBE_CROPS = 20
# this build's energy-VAD rule: mean square of a 30 ms frame above this (int16 LSB^2)
VAD_ENERGY_THRESHOLD = 250000
VAD_FRAME_MS = 30
VAD_PADDING_MS = 300
