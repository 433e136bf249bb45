"""Exposition parser, amd-smi oracle and benchmark helpers."""
