gfalign evalGFA -f testFiles/random1.gfa -g testFiles/random1.gaf --sort-alignment
embedded
+++Alignment summary+++: 
# alignments: 4
Average read length: 91
Average aligned sequence: 37.5
Alignment orientation (+/-): 4(100%):0(0%)
Average path length: 60
Average alignment quality: 60
Average matches #: 37.5
Average block length: 37.5
Primary alignments: 2
Secondary alignments: 2
Supplementary alignments: 2
Terminal supplementary alignments: 0
