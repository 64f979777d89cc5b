"""MI355X-native TransEditor generator hot path."""
__version__ = "0.1.0"
