"""Constants of the reference's lib/blurrily/defaults.rb:2-9."""
DEFAULT_HOST = "localhost"
DEFAULT_PORT = 12021
DEFAULT_DATABASE = "words"

LIMIT_DEFAULT = 10
LIMIT_RANGE = range(1, 1024 + 1)
REF_RANGE = range(1, (1 << 31) + 1)
WEIGHT_RANGE = range(0, (1 << 31) + 1)
