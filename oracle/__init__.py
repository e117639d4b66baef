"""ORACLE package -- test infrastructure only (see oracle_np.py / oracle.c headers)."""
