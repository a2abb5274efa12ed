"""CPU oracle of the hot path: test infrastructure only (see oracle/oracle.py)."""
