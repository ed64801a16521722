// empty stand-in so that the reference's optimizers.cu kernel section compiles on the host (oracle/ref/ref_adam.cpp); test infrastructure only
