"""Signal-processing constants of the reference's model front end
(`/root/reference/constants.py:13-19`), imported as `c` like the reference does."""
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
