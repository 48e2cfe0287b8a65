"""d3h -- MI355X-native hot path of D3-Human's render-and-fit loop (host side of the C ABI in include/d3h.h)."""
