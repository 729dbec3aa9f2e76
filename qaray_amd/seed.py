"""Python mirror of include/qa_seed.h (per-pixel RNG stream seeding)."""

DEFAULT_SEED = 0x51A7A7


def pixel_rand(seed, pixel):
    h = (seed ^ ((pixel * 0x9E3779B9) & 0xFFFFFFFF)) & 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h & 0x7FFFFFFF


def pixel_seed(seed, pixel):
    return pixel_rand(seed, pixel) % 999999999 + 1
