"""Seeded weights / synthetic inputs shared by the oracle, the tests and the benchmark live in cosyvoice_amd/synthetic.py
(they are data, not oracle logic); re-exported here for the oracle-side scripts."""
from cosyvoice_amd.synthetic import *  # noqa: F401,F403
from cosyvoice_amd.synthetic import _Gen  # noqa: F401
