# Behavioural target: Mutation-Simulator 3.0.2 (reference _version.py:1); build tag is ours.
__version__ = "3.0.2"
__amd_build__ = "mi355x-r1"
