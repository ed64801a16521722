/* grt_oracle.c placeholder, filled in below */
