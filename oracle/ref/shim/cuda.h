// empty stand-in so that the kernel section of the reference's particlePrimitives.cu compiles on the host (oracle/ref/ref_grt_proxies.cpp); test infrastructure only
