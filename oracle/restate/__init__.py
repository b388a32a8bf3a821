"""CPU restatements of the reference hot path (test infrastructure, see oracle/__init__.py)."""
