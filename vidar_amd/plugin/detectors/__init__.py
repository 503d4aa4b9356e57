"""vidar_amd.plugin.detectors -- see vidar_amd/plugin/__init__.py for the registry surface."""
