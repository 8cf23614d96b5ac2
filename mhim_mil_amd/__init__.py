"""mhimx — MI355X-native MHIM aggregation path."""
