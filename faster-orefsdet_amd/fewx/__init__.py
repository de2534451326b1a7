"""`fewx` surface of Faster-OreFSDet on the MI355X-native HIP path (new code; same import names as ref:fewx/)."""
