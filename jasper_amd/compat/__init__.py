"""Modules with the names and call shapes of the reference's own Python dependencies, served by libjasper_hip.so."""
