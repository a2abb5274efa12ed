"""Sampling along rays and alpha compositing (render/ray_helper mirror)."""
