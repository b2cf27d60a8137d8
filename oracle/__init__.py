"""Test infrastructure: CPU restatements of the reference's FP4 hot path.

Nothing under torch-bnb-fp4_amd/ may import this package.
"""
