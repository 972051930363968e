"""MI355X-native render path for the lydiasamuel/raytracer scene model (host-side package)."""
