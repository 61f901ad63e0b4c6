"""misti_amd: MI355X-native composite-likelihood engine for MiSTI.

The compute path is the HIP library ``misti_amd/csrc/libmisti_hip.so`` behind the
C ABI declared in ``include/misti_hip.h``; this package holds the Python host
that mirrors the reference's ``MigrationInference`` interface on top of it.
"""
__version__ = "0.1.0"
