"""ORACLE package: CPU restatements of the reference hot path (test infrastructure only)."""
