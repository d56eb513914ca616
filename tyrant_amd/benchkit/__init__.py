"""Parts of bench.py that are not the measurement itself: the child-process modes it runs before it touches the GPU (rocprofv3 --pmc passes,
the instrumented builds' renders, the pre-flight of the native multi-GPU exchange) and the arithmetic that turns counters into the
roofline block.  Test and bench glue like the rest of this package's Python; the product is the C ABI of libtyrant_hip.so."""
