"""Counterpart of the reference's ``model`` package (nets.py, losses.py)."""
