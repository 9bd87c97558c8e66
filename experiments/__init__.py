"""Import-path shim: ``experiments.layers`` / ``experiments.optimized_layers`` resolve to the
MI355X-native drop-ins, so model files written against the reference layout keep working."""
