"""Parity and host-logic tests."""
