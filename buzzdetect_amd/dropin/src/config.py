"""Path and naming constants the plugin loaders resolve against (reference: src/config.py:5-29).
Directories are relative to the working directory, exactly as in the reference."""

DIR_AUDIO = 'audio_in'
SUBDIR_OUTPUT = 'output'

SUFFIX_RESULT_COMPLETE = '_buzzdetect.csv'
SUFFIX_RESULT_PARTIAL = '_buzzpart.csv'
PREFIX_COLUMN_ACTIVATION = 'activation_'
PREFIX_COLUMN_DETECTION = 'detections_'

BAD_READ_ALLOWANCE = 0.01
FILE_SIZE_MINIMUM = 5000

DIR_EMBEDDERS = 'embedders'

DIR_MODELS = 'models'
DEFAULT_MODEL = 'model_general_v3'
SUBDIR_TESTS = 'tests'
FNAME_METRICS = 'metrics.csv'
