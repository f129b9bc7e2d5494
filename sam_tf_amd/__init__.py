"""samx: MI355X-native SAM update step — Python mirror of the reference's agent / discriminator / replay
classes over the C ABI of libsamx.so (include/samx.h).  See DESIGN.md and INTEGRATION.md."""
