"""Counterparts of the reference's ``evaluation`` scripts on the hot path."""
