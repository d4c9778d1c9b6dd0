"""re-export of the oracle backend patch helper (oracle/backend.py) for the tests"""
from oracle.backend import PATCHES, oracle_ops, patch_ops  # noqa: F401
