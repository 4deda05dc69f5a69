"""placeholder, filled in below"""
