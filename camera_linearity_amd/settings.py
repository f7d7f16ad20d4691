"""Explicit settings of the hot path - the subset of the reference's GlobalSettings
(modules/global_settings.py:9-81, read there from data/config.ini at import time) that the merge
path consumes. Here they are plain module-level values with the reference's names; callers override
them per call or through `configure()`. Nothing is read from disk at import.
"""
BIT_DEPTH = 8                  # global_settings.py:35
BITS = 2 ** BIT_DEPTH          # :36
MAX_DN = BITS - 1              # :37
MIN_DN = 0                     # :38
NUM_OF_CHS = 3                 # :30  (B, G, R - OpenCV order, :32)
CH_STR = {0: "Blue", 1: "Green", 2: "Red"}

IM_SIZE_X = None               # :16  image rows; None = take from the data
IM_SIZE_Y = None               # :17  image columns
DARK_THRESHOLD = 0.05          # :62  doubles as exposure gate [s] (image_set.py:173) and pixel threshold (:387)
FF_MID_PERCENTAGE = 0.2        # :63  flat-field ROI fraction
MEDIAN_FILTER_KERNEL_SIZE = 3  # :65
LOWER_LIN_LIM = 5              # :68
UPPER_LIN_LIM = 250            # :69

# Paths behind the reference's default arguments (global_settings.py:40-60). None = not configured: nothing is looked up.
ICRF_CALIBRATED_FILE = None    # text table (BITS, C) read by process_HDR_image(ICRF=None), exposure_series.py:406-407
DEFAULT_DARK_PATH = None       # directory of dark frames, exposure_series.py:409 / image_set.py:170
DEFAULT_FLAT_PATH = None       # directory of flat fields, image_set.py:146-155
DATA_PATH = None               # directory of the text tables read_txt_to_array loads by name (global_settings.py: the data directory)
DATAPOINTS = BITS              # samples per ICRF curve after interpolate_data (global_settings.py: 'final datapoints'; equal to BITS = no resampling)


def configure(**kw):
    """Override settings, e.g. configure(DARK_THRESHOLD=0.012, MEDIAN_FILTER_KERNEL_SIZE=5)."""
    g = globals()
    for k, v in kw.items():
        if k not in g or k.startswith("_") or not k.isupper():
            raise KeyError(f"unknown setting {k}")
        if k in ("BIT_DEPTH", "BITS", "MAX_DN") and v != g[k]:
            raise NotImplementedError("the HIP kernels are built for 8-bit DNs (BITS = 256)")
        g[k] = v
