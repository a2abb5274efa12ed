"""Config, registry and tensor helpers the host-side mirror needs."""
