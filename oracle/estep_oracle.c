/* placeholder so `make -C oracle port` links; replaced by the C restatement of the E-step */
int phmrf_oracle_version(void) { return 0; }
