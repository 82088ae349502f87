"""Seed used by every stage CLI of the reference (constants/metadata.py:1)."""
DEFAULT_SEED = 42
